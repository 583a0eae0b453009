// Probe: how fast can ONE compute unit pull L2-resident data, and how does the chip total scale with the number of CUs pulling?
// Every workgroup streams the same few hundred KB (so after the first pass everything is an L2 hit) either into registers
// (global_load_dwordx4, 8 loads in flight per lane) or straight into LDS (global_load_lds_dwordx4, 8 pieces of 1 KiB in flight per
// wave), as rows of 128 bytes (one cache line per 8 lanes: the access shape of the GEMM operand tiles) or 1 KiB contiguous per wave.
// Prints bytes per clock per CU (s_memtime ticks at 100 MHz are not shader clocks: the rate is reported per microsecond and
// converted with the 2.4 GHz peak clock, so the B/clk figure is a lower bound).
// build: hipcc --offload-arch=gfx950 -O3 -o l2_fetch_probe l2_fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GAS __attribute__((address_space(1)))

template <int MODE>      // 0: registers, 128-byte rows; 1: registers, 1 KiB contiguous; 2: LDS-DMA, 128-byte rows; 3: LDS-DMA contiguous
__global__ __launch_bounds__(256) void k_fetch(const float* __restrict__ buf, float* out, int foot_kb, int reps) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t foot_f = (size_t)foot_kb * 256;      // floats
    float acc = 0.f;
    // a wave-instruction covers 1 KiB: either 8 rows x 128 B (row pitch 2 KiB) or 1 KiB contiguous
    const size_t lane_off = (MODE & 1) ? (size_t)4 * lane : (size_t)(lane >> 3) * 512 + 4 * (lane & 7);
    const size_t inst_step = (MODE & 1) ? 256 : 32;   // floats between consecutive instructions of a wave (rows mode: next 128 B of the same 8 rows)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + 8192u * wid;
    for (int r = 0; r < reps; ++r) {
        size_t base = ((size_t)(blockIdx.x * 4 + wid) * 4096 + (size_t)r * 65536) % foot_f;
        for (int it = 0; it < 16; ++it) {
            if (MODE < 2) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t o = (base + lane_off + u * inst_step) % foot_f;
                    v[u] = *(const float4*)(buf + o);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].w;
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t o = (base + lane_off + u * inst_step) % foot_f;
                    const GAS float* src = (const GAS float*)(buf + o);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(lds0 + 1024u * u) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            base = (base + ((MODE & 1) ? 2048 : 4096)) % foot_f;
        }
    }
    if (MODE >= 2) acc = smem[tid];
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, const float* buf, float* out, int blocks, int foot_kb, int n_cu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 64;
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_fetch<MODE>, dim3(blocks), dim3(256), 32768, 0, buf, out, foot_kb, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double bytes = (double)blocks * 4 * reps * 16 * 8 * 1024.0;
    const int cus = blocks < n_cu ? blocks : n_cu;
    printf("%-28s blocks %5d (%.1f per CU) footprint %5d KB: %8.1f us  %8.1f GB/s  = %5.1f GB/s per CU = %5.1f B/clk/CU at 2.4 GHz\n", name, blocks,
           (double)blocks / n_cu, foot_kb, ms * 1000, bytes / ms / 1e6, bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.4);
}

int main() {
    float *buf, *out;
    hipMalloc(&buf, 64u << 20); hipMemset(buf, 0, 64u << 20); hipMalloc(&out, 4);
    int n_cu = 256;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    for (int foot_kb : {512, 16384}) {
        for (int blocks : {64, 256, 512, 1024, 2048}) {
            run<0>("regs, 128-B rows", buf, out, blocks, foot_kb, n_cu);
            run<1>("regs, 1 KiB contiguous", buf, out, blocks, foot_kb, n_cu);
            run<2>("LDS-DMA, 128-B rows", buf, out, blocks, foot_kb, n_cu);
            run<3>("LDS-DMA, 1 KiB contiguous", buf, out, blocks, foot_kb, n_cu);
        }
    }
    return 0;
}
