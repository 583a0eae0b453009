// Probe: issue rate of LDS-DMA pieces from ONE wave (and from 4 / 8 waves of one workgroup on one CU): is a wave's stream of
// global_load_lds_dwordx4 pipelined, or does each piece wait for the one before?  Variants: M0 saved / written / restored around
// every piece (what glds16 does), M0 written once and left alone, and plain register loads of the same bytes.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_rate_probe glds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define GAS __attribute__((address_space(1)))
template <int MODE, int NP>
__global__ void k_rate(const float* __restrict__ src, long long* out, float* sink) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + 16384u * wid;
    const GAS float* p = (const GAS float*)(src + (size_t)wid * 65536 + 4 * lane);
    float acc = 0.f;
    for (int rep = 0; rep < 3; ++rep) {      // rep 0 warms L2; the last rep is reported
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p + 256 * u), "s"(lds0 + 1024u * (u & 15)) : "memory");
            }
        } else if (MODE == 1) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds0) : "memory");
#pragma unroll
            for (int u = 0; u < NP; ++u) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(p + 256 * u) : "memory");
        } else {
            float4 v[NP];
            const float* q = src + (size_t)wid * 65536 + 4 * lane;
            asm volatile("" : "+v"(q));      // a fresh address every repetition: the loads are not hoisted out of the loop
#pragma unroll
            for (int u = 0; u < NP; ++u) v[u] = *(const float4*)(q + 256 * u);
#pragma unroll
            for (int u = 0; u < NP; ++u) asm volatile("" :: "v"(v[u].x), "v"(v[u].w));      // landed values are consumed
#pragma unroll
            for (int u = 0; u < NP; ++u) acc += v[u].x;
        }
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        if (lane == 0) { out[3 * wid] = t1 - t0; out[3 * wid + 1] = t2 - t0; }
    }
    if (acc == 1.234f) sink[0] = acc + smem[lane];
}
template <int MODE, int NP>
static void run(const char* name, int waves, const float* src, long long* out, float* sink) {
    hipLaunchKernelGGL((k_rate<MODE, NP>), dim3(1), dim3(64 * waves), 16384 * waves, 0, src, out, sink);
    hipDeviceSynchronize();
    long long h[24];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s %d wave(s) x %2d pieces of 1 KiB: issued after %6lld clk, landed after %6lld clk  (%.1f clk / piece / wave, %.1f B/clk/CU)\n", name, waves, NP,
           h[0], h[1], (double)h[1] / NP, (double)waves * NP * 1024 / h[1]);
}
int main() {
    float *src, *sink; long long* out;
    hipMalloc(&src, 8 * 65536 * 4 + (1 << 20)); hipMemset(src, 0, 8 * 65536 * 4 + (1 << 20)); hipMalloc(&sink, 4); hipMalloc(&out, 24 * 8);
    hipFuncSetAttribute((const void*)k_rate<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_rate<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int waves : {1, 4, 8}) {
        run<0, 16>("LDS-DMA, M0 set around each piece", waves, src, out, sink);
        run<1, 16>("LDS-DMA, M0 set once", waves, src, out, sink);
        run<2, 16>("registers (global_load_dwordx4)", waves, src, out, sink);
    }
    return 0;
}
