// Probe: does the sustained fp32 MFMA rate depend on the DATA?  (mfma_probe.hip feeds constants: one value per lane, the same in every
// MFMA.)  v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave, one wave per SIMD on every CU, operands out of 8 + 8 registers
// per lane that were loaded from memory: all equal / random normal / random with random signs and exponents; long launches (ms) and
// launches of ~30 us (the size of the scoring kernels).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_data_probe mfma_data_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(const float* __restrict__ src, float* out, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    float x[8], y[8];
    const float* p = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = p[u]; y[u] = p[8 + u]; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[(u + c) & 7], acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
    if (s == 123.f) out[0] = s;
}
static double run(const float* src, float* out, int iters, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, src, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return (double)reps * 256 * 4 * iters * 32 * 4096.0 / ms / 1e9;      // TFLOP/s
}
int main() {
    const size_t n = (size_t)256 * 256 * 16;
    std::vector<float> h(n);
    float *d[3], *out; (void)hipMalloc(&out, 4);
    srand(7);
    for (int kind = 0; kind < 3; ++kind) {
        for (size_t i = 0; i < n; ++i) {
            const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            h[i] = kind == 0 ? 1.0f : kind == 1 ? (float)(0.1 * g) : (float)(g * exp2((double)(rand() % 16 - 8)));
        }
        (void)hipMalloc(&d[kind], n * 4);
        (void)hipMemcpy(d[kind], h.data(), n * 4, hipMemcpyHostToDevice);
    }
    const char* names[3] = {"all operands 1.0", "normal(0, 0.1)", "normal x 2^[-8, 8)"};
    for (int kind = 0; kind < 3; ++kind) {
        const double lng = run(d[kind], out, 20000, 3);          // ~17 ms per launch
        const double sht = run(d[kind], out, 36, 200);           // ~30 us per launch, back to back
        printf("%-22s long launches %6.1f TFLOP/s | 200 launches of 36 x 32 MFMAs per wave %6.1f TFLOP/s (launch gaps included)\n", names[kind], lng, sht);
    }
    // one short launch after 50 ms of idle
    for (int kind = 1; kind < 2; ++kind) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int r = 0; r < 3; ++r) {
            (void)hipDeviceSynchronize();
            struct timespec ts = {0, 50000000}; nanosleep(&ts, nullptr);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, d[kind], out, 512);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("after 50 ms idle: 512 x 32 MFMAs per wave in %.1f us = %.1f TFLOP/s\n", ms * 1e3, 256.0 * 4 * 512 * 32 * 4096 / ms / 1e9);
        }
    }
    return 0;
}
