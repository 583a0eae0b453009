// Probe: what does it cost to mix the two fp32 MFMA shapes in one wave?  Per iteration 32 x v_mfma_f32_32x32x2_f32 on four
// accumulators (k_score_mt's stage) and, optionally, 4 x v_mfma_f32_16x16x4_f32 on a fifth -- in k_score_mt's order (one behind every
// second group of four), bunched at the end of the iteration, or replaced by a fifth 32x32x2 accumulator fed the same way.
// Reports shader clocks (s_memtime) per iteration; one wave per SIMD on every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_mix_probe mfma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, long long* clk, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[8], y[8];
    const float* p = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = p[u]; y[u] = p[8 + u]; }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[g], y[(g + c) & 7], acc[c], 0, 0, 0);
            if (MODE == 1 && (g & 1) == 0) s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g], y[g], s4, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g], y[g], s4, 0, 0, 0);
        }
        if (MODE == 3) {      // four INDEPENDENT 16x16x4 at the end
            f32x4 t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) t[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g], y[g], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s4 += t[0] + t[1] + t[2] + t[3];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = s4[0] + s4[1] + s4[2] + s4[3];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
    if (s == 123.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
}
template <class K>
static void run(const char* name, K kern, const float* src, float* out, long long* dclk) {
    const int iters = 4000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, out, dclk, iters); (void)hipDeviceSynchronize(); }
    long long c; (void)hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f shader clocks per iteration\n", name, (double)c / iters);
}
int main() {
    const size_t n = (size_t)256 * 256 * 16;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(rand() % 2001 - 1000) * 1e-3f;
    float *d, *out; long long* dclk;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&out, 4); (void)hipMalloc(&dclk, 8);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    run("32 x 32x32x2 (4 accumulators)", k<0>, d, out, dclk);
    run("... + 4 x 16x16x4 behind every second group of four", k<1>, d, out, dclk);
    run("... + 4 x 16x16x4 (one chain) at the end of the iteration", k<2>, d, out, dclk);
    run("... + 4 x 16x16x4 (independent) at the end of the iteration", k<3>, d, out, dclk);
    return 0;
}
