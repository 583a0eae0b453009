// Probe: throughput of "column slab" reads -- each wave reads a 128-byte run from each of NROWS rows that are PITCH bytes apart
// (the access pattern of a GEMM operand walked along the batch dimension) -- as a function of PITCH.
// build: hipcc --offload-arch=gfx950 -O3 -o stride_probe stride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ buf, float* out, int pitch_f, int nrows, int slabs_per_row, int reps) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        const long long slab = (wave + (long long)r * gridDim.x * 4) % slabs_per_row;      // which 128-byte column slab
        const float* base = buf + slab * 32 + 2 * li;
        for (int k0 = 0; k0 < nrows; k0 += 32) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float2*)(base + (size_t)(k0 + 4 * u + lg) * pitch_f);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const int nrows = 256;
    float *buf, *out;
    hipMalloc(&buf, 64u << 20); hipMemset(buf, 0, 64u << 20); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int pitch : {1024, 1152, 2048, 2176, 3072, 3200, 4096, 4224, 6144, 6272, 9216, 9344}) {
        const int pitch_f = pitch / 4, slabs = pitch / 128 > 0 ? (pitch / 128) : 1;
        const int blocks = 2048, reps = 8;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, (const float*)buf, out, pitch_f, nrows, slabs, reps);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)blocks * 4 * reps * nrows * 128.0;
        printf("pitch %5d B: %8.1f us  %8.1f GB/s (footprint %.1f MB)\n", pitch, ms * 1000, bytes / ms / 1e6, (double)nrows * pitch / 1e6);
    }
    return 0;
}
