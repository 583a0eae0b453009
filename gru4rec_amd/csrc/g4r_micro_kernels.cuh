// Row gather / scatter micro-benchmark (north_star: "achieved HBM GB/s on the embedding gather/scatter"; SURVEY 7.3 "a
// batched-rows microbenchmark").  The step kernels gather item rows straight into the operand tiles of their GEMMs and scatter
// them from the owner waves of k_update, so no launch of the training step is a pure gather; these kernels isolate the access
// pattern -- `rows` random rows of `W` floats out of a table far larger than the 256 MiB Infinity Cache -- and replace the
// reference's GpuAdvancedSubtensor1_fast gather (custom_theano_ops.py:505-519) and inc/set_subtensor scatter (gru4rec.py:335-340,
// 428-431) as objects of measurement.  One wave per row, RPW rows in flight per wave, 16-byte loads / stores (a 1 KiB row is one
// dwordx4 per lane).
#pragma once
#include "g4r_device.cuh"

#define MB_RPW 4      // rows a wave keeps in flight

// mode 0: out[j] = table[idx[j]] (read + compact write); mode 1: read only (the consumer-fused gather of the step: a per-row
// checksum is all that leaves the wave); mode 2: Adagrad scatter p -= lr g / sqrt(a + g^2 + eps), a += g^2 on rows idx[j] with the
// gradient rows read from a compact buffer (grad row read + parameter r/w + accumulator r/w = 5 row transfers)
template <int CH>
__global__ __launch_bounds__(256) void k_micro_rows(const float* __restrict__ table_, float* __restrict__ acc_, const int* __restrict__ idx,
                                                    float* __restrict__ buf_, long long rows, int W, int mode) {
    const GAS float* table = (const GAS float*)table_;
    GAS float* tab_w = (GAS float*)table_;
    GAS float* acc = (GAS float*)acc_;
    GAS float* buf = (GAS float*)buf_;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long j0 = wave * MB_RPW;
    if (j0 >= rows) return;
    const int nc4 = W >> 2;
    int it[MB_RPW];
#pragma unroll
    for (int r = 0; r < MB_RPW; ++r) it[r] = idx[min(j0 + r, rows - 1)];
    float4 p[MB_RPW][CH], a[MB_RPW][CH], g[MB_RPW][CH];
#pragma unroll
    for (int r = 0; r < MB_RPW; ++r)
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c = 4 * min(lane + 64 * q, nc4 - 1);
            p[r][q] = ld4(table + (size_t)it[r] * W + c);
            if (mode == 2) {
                a[r][q] = ld4(acc + (size_t)it[r] * W + c);
                g[r][q] = ld4(buf + (size_t)min(j0 + r, rows - 1) * W + c);
            }
        }
#pragma unroll
    for (int r = 0; r < MB_RPW; ++r) {
        if (j0 + r >= rows) break;
        float cs = 0.f;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c4 = lane + 64 * q;
            if (mode == 0) {
                if (c4 < nc4) st4(buf + (size_t)(j0 + r) * W + 4 * c4, p[r][q]);
            } else if (mode == 1) {
                cs += (p[r][q].x + p[r][q].y) + (p[r][q].z + p[r][q].w);
            } else {
                const float4 gg = g[r][q];
                float4 an = a[r][q], pn = p[r][q];
                an.x += gg.x * gg.x; an.y += gg.y * gg.y; an.z += gg.z * gg.z; an.w += gg.w * gg.w;
                pn.x -= 0.05f * gg.x * frsq(an.x + G4R_EPS_ADAGRAD); pn.y -= 0.05f * gg.y * frsq(an.y + G4R_EPS_ADAGRAD);
                pn.z -= 0.05f * gg.z * frsq(an.z + G4R_EPS_ADAGRAD); pn.w -= 0.05f * gg.w * frsq(an.w + G4R_EPS_ADAGRAD);
                if (c4 < nc4) { st4(tab_w + (size_t)it[r] * W + 4 * c4, pn); st4(acc + (size_t)it[r] * W + 4 * c4, an); }
            }
        }
        if (mode == 1) { cs = wave_sum(cs); if (lane == 0) buf[j0 + r] = cs; }
    }
}
template __global__ void k_micro_rows<1>(const float*, float*, const int*, float*, long long, int, int);
template __global__ void k_micro_rows<2>(const float*, float*, const int*, float*, long long, int, int);

// Memory-system load for tests/test_gpu_stress.py: every workgroup streams its 256 KiB slice of a buffer far larger than the
// Infinity Cache (16-byte loads, 16-byte stores of the complemented bits), so that the step kernels running next to it on the
// library's stream see HBM latencies several times the idle ones -- the condition under which round 3's stale-register pipeline
// (mutant 4) committed registers whose loads had not landed, and which no parity test on an idle GPU reproduces.
__global__ __launch_bounds__(256) void k_stress_stream(float* __restrict__ buf_, long long n4) {
    GAS float4* buf = (GAS float4*)buf_;
    constexpr int PER_BLOCK = 16384;      // float4 per workgroup = 256 KiB
    const long long base = (long long)blockIdx.x * PER_BLOCK;
#pragma unroll 4
    for (int i = threadIdx.x; i < PER_BLOCK; i += 256) {
        const long long j = base + i;
        if (j < n4) {
            float4 v = buf[j];
            v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w;
            buf[j] = v;
        }
    }
}
