// Device-side helpers for the gfx950 (MI355X) GRU4Rec kernels: wave64 reductions, fp32 MFMA tile
// primitive, Philox4x32-10, activations.  CDNA4 only (wave = 64 lanes; v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gru4rec_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define G4R_EPS_LOSS 1e-24f    /* gru4rec.py:230,241 */
#define G4R_EPS_ADAGRAD 1e-6f  /* gru4rec.py:330 */

// Philox stream ids (counter word 3); twin of oracle/philox.py
#define G4R_STREAM_SAMPLE 0x53414D50u
#define G4R_STREAM_DROP_EMBED 0x44454D42u
#define G4R_STREAM_DROP_HIDDEN 0x44484944u

// ---------------------------------------------------------------------------------------------
// Device-resident step state.  The step index lives on the device so that a captured hipGraph of
// step kernels is step-agnostic: the first kernel of a step reads *_a and republishes it as *_b,
// the middle kernels read *_b, the last kernel writes *_a = *_b + 1.  No kernel both reads and
// writes the same word, so stream order alone makes it race-free.
struct StepState {
    long long t_a, t_b;   // plan step within the epoch
    long long g_a, g_b;   // global step (drives sample-store row, dropout counters, H ping-pong parity)
    int nan_flag;
    int pad;
};

struct DevModel {
    // ---- configuration
    int n_items, n_layers, B, ns, N, R, ldSc;
    int loss, final_act, hidden_act, embed_mode;
    float fa_p0, fa_p1, ha_p0, ha_p1;
    float lr, mom, lmbd, bpreg, logq, inv_B;
    float drop_h, drop_e;
    unsigned long long seed;
    int D[G4R_MAX_LAYERS], IN[G4R_MAX_LAYERS];
    int Dtop, Ein;
    // ---- dense GRU parameters: one flat buffer [Wx0|Wh0|Wrz0|Bh0|Wx1|...]
    int offWx[G4R_MAX_LAYERS], offWh[G4R_MAX_LAYERS], offWrz[G4R_MAX_LAYERS], offBh[G4R_MAX_LAYERS];
    int dense_count;
    float *dense_p, *dense_acc, *dense_vel, *dense_g;
    int apply_dense_inplace;   // 1: Adagrad fused into the gradient kernel (single GPU)
    float grad_scale;          // 1/nranks when gradients are all-reduced
    // ---- sparse tables (row-major, row = item)
    float *Wy, *accWy, *velWy, *By, *accBy, *velBy, *E, *accE, *velE;
    // ---- per-layer state and saved activations
    float *H[G4R_MAX_LAYERS][2];
    float *r[G4R_MAX_LAYERS], *z[G4R_MAX_LAYERS], *c[G4R_MAX_LAYERS], *hd[G4R_MAX_LAYERS], *Hr[G4R_MAX_LAYERS];
    float *dV[G4R_MAX_LAYERS], *dyl[G4R_MAX_LAYERS];
    float *yin0;
    // ---- scoring / loss
    float *Sc, *dSx, *dSy, *dSBy, *dhpart, *lossrow, *loss_steps;
    int ksplit, kch;
    int *occ_idx;   // [R] item of each gathered-row occurrence (X | Y | samples), -1 = inactive
    int *col_item;  // [ldSc] item of each score column, -1 = inactive
    // ---- plan + samples
    const int *in_idx, *out_idx, *Mplan;
    const unsigned char* reset;
    const int* ST;
    int gl;
    const float *lq_tgt, *lq_smp;
    StepState* st;
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reductions for a 256-thread (4-wave) block; `red` is 8 floats of LDS scratch
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// D(16x16) += A(16x4) * B(4x16), fp32 in / fp32 accumulate (exact fmaf chain).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; holds D[row = 4*(l>>4)+reg][col = l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                 unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 r = {c0, c1, c2, c3};
    return r;
}
__device__ __forceinline__ float u32_to_unit(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// dropout multiplier (0 or 1/retain) for element (row, col) at global step g; twin of oracle.philox.dropout_mask
__device__ __forceinline__ float drop_mult(unsigned long long seed, unsigned g, unsigned stream, int row, int col,
                                           float retain) {
    const Philox4 p = philox4x32_10((unsigned)(col >> 2), (unsigned)row, g, stream, (unsigned)seed,
                                    (unsigned)(seed >> 32));
    const int e = col & 3;
    const unsigned x = e == 0 ? p.x : (e == 1 ? p.y : (e == 2 ? p.z : p.w));
    return u32_to_unit(x) < retain ? 1.0f / retain : 0.0f;
}
__device__ __forceinline__ float4 drop_mult4(unsigned long long seed, unsigned g, unsigned stream, int row, int col4,
                                             float retain) {
    const Philox4 p = philox4x32_10((unsigned)col4, (unsigned)row, g, stream, (unsigned)seed, (unsigned)(seed >> 32));
    const float inv = 1.0f / retain;
    float4 m;
    m.x = u32_to_unit(p.x) < retain ? inv : 0.0f;
    m.y = u32_to_unit(p.y) < retain ? inv : 0.0f;
    m.z = u32_to_unit(p.z) < retain ? inv : 0.0f;
    m.w = u32_to_unit(p.w) < retain ? inv : 0.0f;
    return m;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// element-wise activations, gru4rec.py:189-223
__device__ __forceinline__ float act_fwd(int kind, float p0, float p1, float x) {
    switch (kind) {
        case G4R_ACT_RELU: return fmaxf(x, 0.0f);
        case G4R_ACT_TANH: return tanhf(x);
        case G4R_ACT_LEAKY: return x >= 0.0f ? x : p0 * x;
        case G4R_ACT_ELU: return x >= 0.0f ? x : p0 * (expf(x) - 1.0f);
        case G4R_ACT_SELU: return p0 * (x >= 0.0f ? x : p1 * (expf(x) - 1.0f));
        default: return x;
    }
}
// derivative expressed through the OUTPUT y (sign(y) == sign(x) for all of these)
__device__ __forceinline__ float act_bwd_from_out(int kind, float p0, float p1, float y) {
    switch (kind) {
        case G4R_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case G4R_ACT_TANH: return 1.0f - y * y;
        case G4R_ACT_LEAKY: return y >= 0.0f ? 1.0f : p0;
        case G4R_ACT_ELU: return y >= 0.0f ? 1.0f : y + p0;
        case G4R_ACT_SELU: return y >= 0.0f ? p0 : y + p0 * p1;
        default: return 1.0f;
    }
}
