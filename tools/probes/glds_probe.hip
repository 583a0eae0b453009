// Probe: semantics of global_load_lds_dwordx4 (LDS-DMA) that the kernels rely on or would like to rely on:
//   (a) LDS destination = M0 base + 16 bytes x lane, also for bases above 64 KiB (160 KiB LDS on gfx950);
//   (b) lanes switched off in EXEC write nothing (padded LDS layouts: the pad slots belong to masked lanes);
//   (c) a base that is only 8-byte aligned.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_probe glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GAS __attribute__((address_space(1)))

__device__ __forceinline__ void glds16(const GAS float* src, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}

__global__ __launch_bounds__(64) void k_probe(const float* __restrict__ src, float* out, unsigned base_byte, int nactive) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) smem[i] = -1.f;      // 160 KiB
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    if (lane < nactive) glds16((const GAS float*)(src + 4 * lane), lds0 + base_byte);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: the 64 + 8 quads around the destination
    for (int i = lane; i < 4 * 72; i += 64) out[i] = smem[base_byte / 4 - 16 + i];
}

int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *src, *out;
    hipMalloc(&src, 1024); hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    hipMalloc(&out, 4 * 72 * 4);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct { unsigned base; int nact; const char* what; } cases[] = {
        {4096, 64, "base 4 KiB, all lanes"}, {100 * 1024, 64, "base 100 KiB, all lanes"}, {150 * 1024, 64, "base 150 KiB, all lanes"},
        {8192, 56, "base 8 KiB, 56 lanes active"}, {8192 + 8, 64, "base 8 KiB + 8 bytes"}, {8192 + 4, 64, "base 8 KiB + 4 bytes"}};
    for (auto& c : cases) {
        hipMemset(out, 0, 4 * 72 * 4);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 160 * 1024, 0, (const float*)src, out, c.base, c.nact);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> r(4 * 72);
        hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
        int ok = 0, untouched = 0;
        for (int i = 0; i < 256; ++i) { if (r[16 + i] == (float)i) ++ok; if (r[16 + i] == -1.f) ++untouched; }
        printf("%-32s: %s | floats landed in place %3d / 256, left untouched %3d | before: %g %g  first: %g %g %g %g  after: %g %g\n", c.what,
               hipGetErrorString(e), ok, untouched, r[14], r[15], r[16], r[17], r[18], r[19], r[16 + 256], r[16 + 257]);
    }
    return 0;
}
