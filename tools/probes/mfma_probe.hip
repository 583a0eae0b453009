// Probe: sustained rate of the fp32 MFMA shapes with register-resident operands (N independent accumulator chains per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 123.f) out[0] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
    if (s == 123.f) out[0] = s;
}
template <class K>
void run(const char* name, K kern, double flop_per_inst, int ch, int waves_per_simd) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000, blocks = 256 * waves_per_simd;      // 256 threads = 4 waves = 1 per SIMD per block
    for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * iters * ch * flop_per_inst;
    printf("%-28s chains %d  waves/SIMD %d : %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ch, waves_per_simd, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)iters * ch * waves_per_simd));
}
int main() {
    run("v_mfma_f32_16x16x4_f32", k16<1>, 2048, 1, 1);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 2048, 4, 1);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 2048, 4, 2);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 2048, 4, 4);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 4096, 1, 1);
    run("v_mfma_f32_32x32x2_f32", k32<2>, 4096, 2, 1);
    run("v_mfma_f32_32x32x2_f32", k32<2>, 4096, 2, 2);
    run("v_mfma_f32_32x32x2_f32", k32<2>, 4096, 2, 4);
    return 0;
}
