// Training / prediction step kernels (gfx950).  Every GEMM of the step is an LDS-staged fp32 MFMA tile GEMM, with gathers,
// dropout, gates and optimizer updates fused into the operand providers / epilogues.  One training step (reference: the
// Theano function built at gru4rec.py:572-584 and called at :623) for one GRU layer of up to 112 units is 6 launches:
//   k_gru_fwd_fused  [y | H] rows -> r, z, candidate, h, next H in one launch (gather + dropout fused)     gru4rec.py:438-479
//   k_score_fwd      Sc = h Wy[Y | samples]^T + By - logq lq            gathered rows through LDS          gru4rec.py:480-495
//   k_loss_rows      final activation + loss + d cost / d s per row                                        gru4rec.py:193-248,496
//   k_score_bwd      dSy = ds^T h, dSBy ; split-K slabs of dh = ds Sy                                      (T.grad, :383-384)
//   k_gru_bwd_fused  slab sum + gate derivatives, dr' GEMM, dy GEMM -> dSx / the lower layer's dh          (T.grad)
//   k_update         dense-gradient tiles (dWx / dWh / dWrz / dBh + dense Adagrad, gru4rec.py:390-406), per-occurrence sparse
//                    Adagrad on the touched Wy / By / E rows (:407-431), step bookkeeping + staging of the next step's inputs
// Wider layers, one-hot input and prediction use the unfused kernels (g4r_gemm.cuh tiles over >= 100 workgroups):
//   k_gru_p1 (V = [y | H] [Wx ; Wrz] + Bh, r, z, H*r), k_gru_p2 (candidate, h), k_gru_bwd_pre / _a / _b, k_onehot_step;
// N > 1 and the generic optimizers: k_dense_grad / k_dense_apply, k_sparse_update(_generic), k_grad_sqsum / k_grad_clip.
#pragma once
#include <type_traits>

#include "g4r_gemm.cuh"

struct StepCtx { long long t, g; int M; };

// `st` is passed to every step kernel as a kernel argument (not read through the descriptor), so these loads
// are issued together with the descriptor-field loads: one round trip to know (t, g, M).
__device__ __forceinline__ StepCtx load_ctx_first(StepState* st_) {
    GAS StepState* st = (GAS StepState*)st_;
    StepCtx c;
    c.t = st->t_a; c.g = st->g_a; c.M = st->M_a;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { st->t_b = c.t; st->g_b = c.g; st->M_b = c.M; }
    return c;
}
__device__ __forceinline__ StepCtx load_ctx(StepState* st_) {
    const GAS StepState* st = (const GAS StepState*)st_;
    StepCtx c;
    c.t = st->t_b; c.g = st->g_b; c.M = st->M_b;
    return c;
}

// Stages the inputs of step (t, g) with M active rows: its in_idx row and its column -> item list (gru4rec.py:436-437: the
// targets of the active rows, then this step's row of the negative-sample store).  All threads of the calling workgroup.
// M = 0 marks a padding step (ranks of a data-parallel run have plans of different lengths): no column is active, so the step
// computes zero dense gradients and touches no item row, but still takes part in the all-reduce.
__device__ __forceinline__ void stage_step_inputs(const DevModel& m, long long t, long long g, int M, int tid, int nth) {
    const int B = m.B, N = m.N, ld = m.ldSc;
    const GAS int* in = m.in_idx + t * B;
    const GAS int* out = m.out_idx + t * B;
    const GAS int* smp = m.ST + (size_t)(m.gl > 0 ? g % m.gl : 0) * m.ns;
    GAS int *ci = m.cur_in, *cc = m.cur_col;
    const GAS unsigned char* rst = m.reset + t * B;      // (the plan carries one trailing, zeroed row)
    for (int b = tid; b < B; b += nth) { ci[b] = in[b]; ci[B + b] = rst[b]; }      // cur_in[B ..]: the step's reset flags (k_gru_h)
    // cur_in[2 B ..]: the step's (t, g, M) once more, for kernels that take them with their first VECTOR loads: hipcc sinks a scalar
    // load of the step state to its first use (behind the branches in front of it), where its 0.5 us of latency -- the state is
    // rewritten every step -- is fully exposed (g4r_lean_kernels.cuh)
    if (tid == 0) { ci[2 * B] = (int)(unsigned)g; ci[2 * B + 1] = (int)(g >> 32); ci[2 * B + 2] = M; ci[2 * B + 3] = (int)(unsigned)t; ci[2 * B + 4] = (int)(t >> 32); }
    // 8 columns per thread and pass, all loads of a pass in flight together (clamped addresses, selects afterwards)
    for (int base = 0; base < ld; base += 8 * nth) {
        int vo[8], vs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = base + q * nth + tid;
            vo[q] = out[min(n, B - 1)];
            vs[q] = (m.ns > 0) ? smp[min(max(n - B, 0), m.ns - 1)] : -1;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = base + q * nth + tid;
            if (n < ld) cc[n] = (n < M) ? vo[q] : (n >= B && n < N && M > 0) ? vs[q] : -1;      // M = 0: padding step of a multi-rank plan, nothing is touched
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Arguments of the GRU kernels in prediction mode (train = 0; gru4rec.py:433 predict=True): explicit state
// instead of the device step state, no dropout, no reset switch, nothing saved for a backward pass.
struct GruFwdPredict {
    GP(const int) in_idx;    // layer 0 gather indices
    GP(const float) ysrc;    // layer > 0 input rows
    GP(const float) Hcur;
    GP(float) Hnext;
    GP(float) hout;          // [rows][D]
    GP(float) Vc; GP(float) z; GP(float) Hr;   // scratch [rows][D]
    int M;
};


// Threads per GEMM workgroup.  Kernels whose grid is a few dozen tiles (one workgroup per CU, most CUs idle) run 8 waves:
// two wave groups split each K chunk (g4r_gemm.cuh) so that two waves per SIMD overlap their issue / MFMA latencies
// (measured: k_gru_p1 11.2 -> 8.9 us, k_gru_bwd_b 7.8 -> 6.6, k_dense_grad 5.5 -> 5.1).  The scoring kernels already
// have more workgroups than CUs and are faster with 4 waves.
#define GT_NTH_FEW 512
#define GT_NTH 256
#define GT_BM 32
#define GT_BN 32
#define GT_BK 128
#ifndef P1_BK
#define P1_BK 256     // K = IN + D of GRU phase 1 in one chunk up to IN + D = 256
#define BB_BK 320     // K = 3D of dy = dV Wx^T in one chunk up to D = 106
#endif

#include "g4r_fwd_kernels.cuh"
#include "g4r_score_mt.cuh"
#include "g4r_loss_kernel.cuh"
#include "g4r_bwd_kernels.cuh"
#include "g4r_score_bmt.cuh"
#include "g4r_update_kernels.cuh"
#include "g4r_lean_kernels.cuh"

// ---------------------------------------------------------------------------------------------
// Negative-sample store refill: ST[e] = upper_bound(P, u_e) with the end clamps of the reference's
// GpuBinarySearchSorted (custom_theano_ops.py:318-349); uniforms from Philox (one call per 4 samples).
__global__ __launch_bounds__(256) void k_sample_refill(int* ST, long long n, const float* P, int n_items,
                                                       unsigned long long seed, unsigned refill_no) {
    const long long cidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (cidx * 4 >= n) return;
    const Philox4 p = philox4x32_10((unsigned)cidx, refill_no, 0u, G4R_STREAM_SAMPLE, (unsigned)seed,
                                    (unsigned)(seed >> 32));
    const unsigned xs[4] = {p.x, p.y, p.z, p.w};
    const float minv = P[0], maxv = P[n_items - 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long idx = cidx * 4 + e;
        if (idx >= n) break;
        const float val = u32_to_unit(xs[e]);
        long long a = 0, b = n_items - 1;
        if (val > maxv) { a = n_items; b = n_items; }
        else if (val <= minv) { a = 0; b = 0; }
        while (b - a > 0) {
            const long long hmid = (a + b) / 2;
            if (val < P[hmid]) b = hmid; else a = hmid + 1;
        }
        ST[idx] = (int)b;
    }
}

// hidden-state row compaction (gru4rec.py:647-651): dst[j] = src[map[j]] (map < 0 -> zeros)
__global__ __launch_bounds__(256) void k_gather_rows(float* dst, const float* src, const int* map, int nrows, int W) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * W) return;
    const int j = e / W, d = e - j * W, s = map[j];
    dst[e] = s >= 0 ? src[(size_t)s * W + d] : 0.f;
}

// host-side (re)positioning of the step state: plan step t, global step g; stages that step's inputs (one 512-thread workgroup)
__global__ __launch_bounds__(512) void k_set_state(const DevModel* __restrict__ mp, StepState* st, long long t, long long g) {
    const DevModel& m = *mp;
    const int M = m.Mplan[t];
    if (threadIdx.x == 0) {
        st->t_a = t; st->t_b = t; st->g_a = g; st->g_b = g;
        st->M_a = M; st->M_b = M;
    }
    stage_step_inputs(m, t, g, M, threadIdx.x, 512);
}
// after a sample-store refill between two steps: the staged column list of the step about to run holds the old samples
__global__ __launch_bounds__(512) void k_restage_inputs(const DevModel* __restrict__ mp, StepState* st) {
    const GAS StepState* sg = (const GAS StepState*)st;
    stage_step_inputs(*mp, sg->t_a, sg->g_a, sg->M_a, threadIdx.x, 512);
}
