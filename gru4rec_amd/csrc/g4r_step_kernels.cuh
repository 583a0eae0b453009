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
    for (int b = tid; b < B; b += nth) ci[b] = in[b];
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

// ---------------------------------------------------------------------------------------------
// GRU phase 1: V[B, 3D] = [y | H] * [Wx ; 0|Wrz] + Bh over 32x32 tiles, K = IN + D.
// Epilogue per column block: [0,D) -> Vc (candidate pre-activation part), [D,2D) -> r = sigmoid, Hr = H*r,
// [2D,3D) -> z = sigmoid.  For layer 0 the A provider gathers Wy[X] / E[X] rows and applies embedding dropout.
template <int TBN, int TBK>
__global__ __launch_bounds__(GT_NTH_FEW) void k_gru_p1(const DevModel* __restrict__ mp, StepState* st, int l, int train, int first, GruFwdPredict pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int tid = threadIdx.x;
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D, K = IN + D;
    long long g = 0;
    int M;
    const GAS float *Hcur, *ysrc = nullptr;
    const GAS int* gidx = nullptr;
    GAS float *Vc, *zb, *Hrb, *rb = nullptr;
    if (train) {
        const StepCtx c = first ? load_ctx_first(st) : load_ctx(st);
        g = c.g; M = c.M;
        Hcur = m.H[l][g & 1];
        if (l == 0) gidx = m.cur_in; else ysrc = m.hd[l - 1];      // staged by the previous step's bookkeeping: no wait for t
        Vc = m.Vc[l]; zb = m.z[l]; Hrb = m.Hr[l]; rb = m.r[l];
    } else {
        M = pa.M; Hcur = pa.Hcur; gidx = pa.in_idx; ysrc = pa.ysrc; Vc = pa.Vc; zb = pa.z; Hrb = pa.Hr;
    }
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * TBN;
    GAS long long* clk = (G4R_DBGCLK(m) && blockIdx.x == 1 && blockIdx.y == 1) ? G4R_DBGCLK(m) + 0 : nullptr;      // kernel 0 of tools/clk.py
    if (clk && tid == 0) clk[4] = wall_clock64();     // context known
    // gather indices of the tile's rows go to LDS first: the row loads must not chain behind index loads
    int* sRow = reinterpret_cast<int*>(smem + TileCfg<GT_BM, TBN, TBK, false, false>::SMEM_FLOATS);
    if (tid < GT_BM) {
        const int row = m0 + tid;
        const int item = (l == 0 && row < M) ? gidx[row] : -1;
        sRow[tid] = item;
        if (train && l == 0 && blockIdx.x == 0 && row < m.B) {
            m.occ_idx[row] = item;
            if (item >= 0 && m.xmode == 0) {      // first / last occurrence of the item in this step's gathered-row list (k_sparse_update); exact-replica mode: k_exact_occ publishes the exchanged list instead
                int* fl = (int*)m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED ? 0 : (size_t)m.n_items) + item);
                atomicMax(fl, row + 1);
                atomicMax(fl + 1, m.R - row);
                atomicAdd(fl + 2, 1);
            }
        }
    }
    if (m0 >= M) return;
    __syncthreads();
    const GAS float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const bool onehot = (l == 0 && m.embed_mode == G4R_EMBED_ONEHOT);    // V = Wx[0][X] + Bh + (0 | H Wrz), gru4rec.py:458-460
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wrz = m.dense_p + m.offWrz[l];
    const GAS float* Bh = m.dense_p + m.offBh[l];
    const float retain_e = 1.0f - m.drop_e;
    const float drop_e = m.drop_e;
    const unsigned long long seed = m.seed;
    // raw loads from clamped addresses, no selects or branches between them; zeroing of the out-of-range part and the
    // embedding dropout happen in afix / bfix on the way to LDS (g4r_gemm.cuh: stage_commit)
    auto aload = [&](int kk, int r, int c) -> float4 {
        const int k = min(kk + c, K - 4);
        const bool isy = k < IN;
        const int rowc = min(m0 + r, max(M - 1, 0));      // rows past the batch must not even form an out-of-range address
        const GAS float* src = isy ? ((l == 0) ? table + (size_t)max(sRow[r], 0) * IN : ysrc + (size_t)rowc * IN)
                                   : Hcur + (size_t)rowc * D;
        return ld4(src + (isy ? k : k - IN));
    };
    auto afix = [&](int kk, int r, int c, float4 v) -> float4 {
        const int row = m0 + r, k = kk + c;
        if (!(row < M && k < K)) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (train && l == 0 && drop_e > 0.f && k < IN) {
            const float4 mk = drop_mult4(seed, (unsigned)g, G4R_STREAM_DROP_EMBED, row, k >> 2, retain_e);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        }
        return v;
    };
    auto bload = [&](int kk, int r, int c) -> float4 {
        const int k = min(kk + r, K - 1), n = min(n0 + c, D3 - 4);
        const bool isx = k < IN;
        return ld4(isx ? Wx + (size_t)k * D3 + n : Wrz + (size_t)(k - IN) * (2 * D) + max(n - D, 0));
    };
    auto bfix = [&](int kk, int r, int c, float4 v) -> float4 {
        const int k = kk + r, n = n0 + c;
        const bool ok = k < K && n < D3 && (k < IN || n >= D);
        return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto pre = [&](int row, int n) -> float4 {      // bias and (for the r block) the hidden value
        const bool ok = row < M && n < D3;
        const float oh = onehot ? ldf_at(table, (size_t)max(sRow[row - m0], 0) * D3 + n, ok) : 0.f;
        return make_float4(ldf_at(Bh, n, ok), ldf_at(Hcur, (size_t)row * D + (n - D), ok && n >= D && n < 2 * D), oh, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M || n >= D3) return;
        v += p.x + p.z;
        if (n < D) { Vc[(size_t)row * D + n] = v; return; }
        if (n < 2 * D) {
            const size_t o = (size_t)row * D + (n - D);
            const float rr = sigmoidf_(v);
            if (train) rb[o] = rr;
            Hrb[o] = p.y * rr;
            return;
        }
        zb[(size_t)row * D + (n - 2 * D)] = sigmoidf_(v);
    };
    if (clk && tid == 0) clk[5] = wall_clock64();     // row indices in LDS
    // the first column tile of every row block also publishes its gathered (and dropout-masked) layer-0 input rows: the
    // dense-gradient tiles read them back (dWx = yin^T dV) while the sparse update is already rewriting the table rows
    GAS float* yin0 = m.yin0;
    const bool pub = train && l == 0 && blockIdx.x == 0 && IN > 0;
    auto hook = [&](const float* sA, int kk, int kend) {
        if (!pub) return;
        constexpr int LDA = TileCfg<GT_BM, TBN, TBK, false, false>::LDA;
        const int kmax = min(kend, IN - kk);          // columns of this chunk that belong to y
        const int r = tid >> 4;                       // 32 rows x 16 column slots per pass (no integer division)
        if (m0 + r < M) {
            for (int k = tid & 15; k < kmax; k += 16) yin0[(size_t)(m0 + r) * IN + kk + k] = sA[r * LDA + k];
        }
    };
    gemm_tile<GT_BM, TBN, TBK, false, false, GT_NTH_FEW>(m0, n0, K, aload, bload, pre, epi, smem, clk, hook, afix, bfix);
}

// GRU phase 2: c = act(Hr * Wh + Vc) ; h = (1 - z) H + z c ; hidden dropout ; reset switch (gru4rec.py:474-479)
// NTH / BK: 4 waves and 128-deep chunks where the launch fills the chip; 8 waves (two wave groups that split every chunk's k range) and
// 256-deep chunks where it does not -- there the tile waits out one memory round trip per chunk and one MFMA chain per k-step.
template <int NTH, int BK>
__global__ __launch_bounds__(NTH) void k_gru_p2(const DevModel* __restrict__ mp, StepState* st, int l, int train, GruFwdPredict pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int D = m.D[l];
    long long t = 0, g = 0;
    int M;
    const GAS float *Hcur, *Vc, *zb, *Hrb;
    GAS float *Hnext, *hout;
    if (train) {
        const StepCtx c = load_ctx(st);
        t = c.t; g = c.g; M = c.M;
        Hcur = m.H[l][g & 1]; Hnext = m.H[l][(g + 1) & 1]; hout = m.hd[l];
        Vc = m.Vc[l]; zb = m.z[l]; Hrb = m.Hr[l];
    } else {
        M = pa.M; Hcur = pa.Hcur; Hnext = pa.Hnext; hout = pa.hout; Vc = pa.Vc; zb = pa.z; Hrb = pa.Hr;
    }
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    if (m0 >= M) return;
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS unsigned char* rst = train ? m.reset + t * m.B : nullptr;
    const float retain_h = 1.0f - m.drop_h, drop_h = m.drop_h, hp0 = m.ha_p0, hp1 = m.ha_p1;
    const int hact = m.hidden_act;
    const unsigned long long seed = m.seed;
    GAS float* cl = m.c[l];
    auto aload = [&](int kk, int r, int c) -> float4 {
        const int row = m0 + r, k = kk + c;
        return ld4_if(Hrb, (size_t)row * D + k, row < M && k < D);
    };
    auto bload = [&](int kk, int r, int c) -> float4 {
        const int k = kk + r, n = n0 + c;
        return ld4_if(Wh, (size_t)k * D + n, k < D && n < D);
    };
    auto pre = [&](int row, int n) -> float4 {
        const bool ok = row < M && n < D;
        const size_t o = (size_t)row * D + n;
        float4 p = make_float4(ldf_at(Vc, o, ok), ldf_at(zb, o, ok), ldf_at(Hcur, o, ok), 0.f);
        if (train) p.w = rst[ok ? row : 0] ? 1.f : 0.f;
        return p;
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M || n >= D) return;
        const size_t o = (size_t)row * D + n;
        const float cc = act_fwd(hact, hp0, hp1, v + p.x);
        const float zz = p.y;
        float h = (1.0f - zz) * p.z + zz * cc;
        if (train) {
            if (drop_h > 0.f) h *= drop_mult(seed, (unsigned)g, G4R_STREAM_DROP_HIDDEN + l, row, n, retain_h);
            cl[o] = cc;
            hout[o] = h;
            Hnext[o] = p.w != 0.f ? 0.f : h;
        } else {
            hout[o] = h;
            Hnext[o] = h;
        }
    };
    gemm_tile<GT_BM, GT_BN, BK, false, false, NTH>(m0, n0, D, aload, bload, pre, epi, smem);
}

// ---------------------------------------------------------------------------------------------
// GRU forward of one layer in ONE launch (training, layers whose operands fit the LDS plan below: in + D <= ~200): replaces
// k_gru_p1 + k_gru_p2.  One 8-wave workgroup per 16 rows x 32 output columns.  The candidate needs (H * r) Wh over ALL D
// columns, so every column tile computes r for all D columns of its rows (a 16 x D x (in + D) product, repeated by the four
// column tiles of a row block: cheaper than a launch boundary), z and the candidate's input part only for its own 32 columns:
//   stage A1  K = input part (k < in):  V_r (D cols), V_z (32), V_c (32)      operands [y | H] rows, Wx column blocks in LDS
//   stage A2  K = hidden part:          V_r, V_z += H * Wrz                   (the V_r weight buffer is reused)
//   epilogue  r = sigmoid, Hr = H r -> LDS (+ memory for the tile's own columns); z, V_c -> LDS
//   stage B   (H r) Wh for the 32 columns (two sub-tiles x four quarters of K over the eight waves), joined through LDS
//   epilogue  c = act(.), h = (1 - z) H + z c, hidden dropout, reset switch -> H_next; saves c, hd   (gru4rec.py:471-479)
// Layer 0 gathers its input rows (+ embedding dropout), publishes them (yin0) and the X part of occ_idx / occ_fl, and copies
// the step state, exactly as k_gru_p1 does.  B operands are kept [k][n] with row strides == 16 mod 32 (conflict-free reads).
#define FF_ROWS 16
#define FF_LDR 112      // row stride of the V_r weight buffer (D <= 112)
#define FF_LDT 48       // row stride of the 32-column weight tiles
struct FwdFusedLds {     // float offsets of the LDS plan
    int sA, sWr, sWz, sWc, sWh, sHr, sZ, sVc, sRow, sJoin, total;
    int LDA, LDH;
};
__host__ __device__ inline FwdFusedLds fwd_fused_lds(int IN, int D) {
    FwdFusedLds o;
    const int KA = IN + D, rk = IN > D ? IN : D;
    o.LDA = KA + 2; o.LDH = D + 2;
    o.sA = 0;
    o.sWr = o.sA + FF_ROWS * o.LDA;
    o.sWz = o.sWr + rk * FF_LDR;
    o.sWc = o.sWz + KA * FF_LDT;
    o.sWh = o.sWc + IN * FF_LDT;
    o.sHr = o.sWh + D * FF_LDT;
    o.sZ = o.sHr + FF_ROWS * o.LDH;
    o.sVc = o.sZ + FF_ROWS * 33;
    o.sRow = o.sVc + FF_ROWS * 33;
    o.sJoin = (o.sRow + FF_ROWS + 3) & ~3;          // [6][64] f32x4 partial sums of stage B
    o.total = o.sJoin + 6 * 64 * 4;
    return o;
}
__global__ __launch_bounds__(512) void k_gru_fwd_fused(const DevModel* __restrict__ mp, StepState* st, int l, int first) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D, D2 = 2 * D, KA = IN + D, Dq = D >> 2, INq = IN >> 2;
    // step context: the loads are issued here, the values are first USED behind the requests that do not depend on them (weights
    // by LDS-DMA and registers, row items, biases) -- a use up here would put the state's memory round trip (the previous launch
    // wrote it) in front of everything
    const GAS StepState* sgc = (const GAS StepState*)st;
    const long long t = first ? sgc->t_a : sgc->t_b, g = first ? sgc->g_a : sgc->g_b;
    const int M = first ? sgc->M_a : sgc->M_b, B = m.B;
    const int m0 = blockIdx.y * FF_ROWS, n0 = blockIdx.x * 32;
    const FwdFusedLds L = fwd_fused_lds(IN, D);
    float* sA = smem + L.sA;       // [16][LDA]   [y | H] rows
    float* sWr = smem + L.sWr;     // [max(in, D)][FF_LDR]   Wx[:, D:2D], then Wrz[:, 0:D]
    float* sWz = smem + L.sWz;     // [in + D][FF_LDT]       [Wx[:, 2D + n0 ..] ; Wrz[:, D + n0 ..]]
    float* sWc = smem + L.sWc;     // [in][FF_LDT]           Wx[:, n0 ..]
    float* sWh = smem + L.sWh;     // [D][FF_LDT]            Wh[:, n0 ..]
    float* sHr = smem + L.sHr;     // [16][LDH]
    float* sZ = smem + L.sZ;       // [16][33]
    float* sVc = smem + L.sVc;     // [16][33]
    int* sRow = reinterpret_cast<int*>(smem + L.sRow);
    f32x4* sJ = reinterpret_cast<f32x4*>(smem + L.sJoin);
    const int LDA = L.LDA, LDH = L.LDH;
    GAS long long* clk = (G4R_DBGCLK(m) && blockIdx.x == 1 && blockIdx.y == 1) ? G4R_DBGCLK(m) + 0 : nullptr;      // kernel 0 of tools/clk.py
    if (clk && tid == 0) clk[0] = wall_clock64();
    // ---- row items first (the gathers wait for them), then everything that does not depend on them
    const int rrow = m0 + (tid & 15);
    int item = (l == 0) ? m.cur_in[min(rrow, B - 1)] : 0;      // staged by the previous step's bookkeeping: no wait for t
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wrz = m.dense_p + m.offWrz[l];
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float* Bh = m.dense_p + m.offBh[l];
    // V_r weights: 16 rows of k per pass, one quad of n per thread (32 quad slots, Dq <= 28 used)
    constexpr int NP_R = 7;
    const int kr = tid >> 5, nq = min(tid & 31, Dq - 1);
    float4 wr1[NP_R];
#pragma unroll
    for (int p = 0; p < NP_R; ++p) wr1[p] = ld4(Wx + (size_t)min(kr + 16 * p, IN - 1) * D3 + D + 4 * nq);
    // Everything else that does not wait for the gather goes global -> LDS by LDS-DMA, into the padded [k][n] tiles (~90 KB per
    // workgroup without passing through registers: 14 + 8 quads per thread less to hold and to store; k_gru_fwd_fused 10.5 -> 10.2 us
    // at D = 100 -- the phase is bound by the first-touch latency of weights another XCD rewrote a few microseconds ago, not by the
    // copy).  Columns of the 32-column tiles past the matrix edge read clamped addresses: they only feed output columns that are
    // never stored.
    {      // (unconditional: a row block past the batch waits for its pieces before it leaves, below)
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        dma_rows<FF_LDR / 4, 8>(lds0 + 4u * L.sWr, D, Dq, wid, lane, [&](int k, int q) { return Wrz + (size_t)k * D2 + 4 * q; });
        dma_rows<FF_LDT / 4, 8>(lds0 + 4u * L.sWz, KA, 8, wid, lane, [&](int k, int q) {
            const int nzq = min(n0 + 4 * q, D - 4);
            return k < IN ? Wx + (size_t)k * D3 + D2 + nzq : Wrz + (size_t)(k - IN) * D2 + D + nzq;
        });
        dma_rows<FF_LDT / 4, 8>(lds0 + 4u * L.sWc, IN, 8, wid, lane, [&](int k, int q) { return Wx + (size_t)k * D3 + min(n0 + 4 * q, D - 4); });
        dma_rows<FF_LDT / 4, 8>(lds0 + 4u * L.sWh, D, 8, wid, lane, [&](int k, int q) { return Wh + (size_t)k * D + min(n0 + 4 * q, D - 4); });
    }
    // epilogue operands of this wave's sub-tiles: biases of the r columns (16 wid + li), of the tile's z / c columns
    const int nr = wid * 16 + li;
    const float b_r = ldf_at(Bh, D + nr, nr < D);
    const int nt = n0 + (wid & 1) * 16 + li;
    const float b_z = ldf_at(Bh, D2 + nt, nt < D), b_c = ldf_at(Bh, nt, nt < D);
    // ---- first uses of the step context
    if (first && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { GAS StepState* sw = (GAS StepState*)st; sw->t_b = t; sw->g_b = g; sw->M_b = M; }
    if (!(l == 0 && rrow < M)) item = -1;
    const GAS float* Hcur = m.H[l][g & 1];
    // hidden part of the A rows: 16 rows x 32 quad slots
    const int ar = tid >> 5, aq = tid & 31;
    const int arow = min(m0 + ar, max(M - 1, 0));
    const float4 ah = ld4(Hcur + (size_t)max(arow, 0) * D + 4 * min(aq, Dq - 1));
    unsigned rst4 = 0;      // reset flags of rows 4 lg .. 4 lg + 3 (stage-B epilogue: waves 0 and 1)
    if (wid < 2) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rst4 |= (unsigned)m.reset[t * B + min(m0 + 4 * lg + rg, B - 1)] << (8 * rg);
    }
    if (clk && tid == 0) clk[1] = wall_clock64();
    if (tid < FF_ROWS) {
        sRow[tid] = item;
        if (l == 0 && blockIdx.x == 0 && rrow < B) {
            m.occ_idx[rrow] = item;
            if (item >= 0 && m.xmode == 0) {      // first / last occurrence of the item in this step's gathered-row list (k_update)
                int* fl = (int*)m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED ? 0 : (size_t)m.n_items) + item);
                atomicMax(fl, rrow + 1);
                atomicMax(fl + 1, m.R - rrow);
                atomicAdd(fl + 2, 1);
            }
        }
    }
    if (m0 >= M) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }      // its DMA pieces must not land in a later workgroup's LDS
    // ---- everything that does not wait for the gather goes to LDS now ([k][n] tiles, 16-byte stores): the hidden-part
    // weights of V_r (the input part follows into the same buffer after stage A1), the 32-column tiles, the H part of the rows
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool arow_ok = m0 + ar < M;      // rows past the batch are zero
    if (aq < Dq) {      // row stride == 2 mod 4: 8-byte stores
        float2* d = reinterpret_cast<float2*>(sA + ar * LDA + IN + 4 * aq);
        d[0] = arow_ok ? make_float2(ah.x, ah.y) : make_float2(0.f, 0.f);
        d[1] = arow_ok ? make_float2(ah.z, ah.w) : make_float2(0.f, 0.f);
    }
    if (clk && tid == 0) clk[2] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed (hipcc does not count them)
    __syncthreads();
    if (clk && tid == 0) clk[3] = wall_clock64();
    // input part of the A rows: gathered table rows (layer 0) or the lower layer's output; in flight during stage A1
    const GAS float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const GAS float* ysrc = (l == 0) ? table + (size_t)max(sRow[ar], 0) * IN : m.hd[l - 1] + (size_t)max(arow, 0) * IN;
    float4 ay = ld4(ysrc + 4 * min(aq, INq - 1));
    // k-steps in fully unrolled groups of 8 (fragment reads ahead of the MFMAs), single steps for the remainder
    auto mma = [&](f32x4 acc, const float* pa, const float* pb, int ldb, int nk) -> f32x4 {      // pa[k], pb[k * ldb], k = 0, 4, .. < nk
        int k0 = 0;
        for (; k0 + 32 <= nk; k0 += 32) {
            float af[8], bf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { af[u] = pa[k0 + 4 * u]; bf[u] = pb[(k0 + 4 * u) * ldb]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma16(af[u], bf[u], acc);
        }
        for (; k0 < nk; k0 += 4) acc = mfma16(pa[k0], pb[k0 * ldb], acc);
        return acc;
    };
    // ---- stage A.  Wave w < NT1 owns r sub-tile w; the z / c sub-tiles of the tile go to the waves 7, 6 (z) and 5, 4 (c).
    // A1: K = hidden part (needs nothing from the gather), A2: K = input part
    const int NT1 = (D + 15) >> 4;
    const float* paA = sA + li * LDA + lg;
    f32x4 accR = (f32x4){0.f, 0.f, 0.f, 0.f}, accT = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool doZ = wid >= 6, doC = (wid == 4 || wid == 5);
    const int tsub = wid & 1;
    if (wid < NT1) accR = mma(accR, paA + IN, sWr + lg * FF_LDR + wid * 16 + li, FF_LDR, D);
    if (doZ) accT = mma(accT, paA + IN, sWz + (IN + lg) * FF_LDT + tsub * 16 + li, FF_LDT, D);
    if (clk && tid == 0) clk[4] = wall_clock64();
    __syncthreads();
    if (clk && tid == 0) clk[5] = wall_clock64();
#pragma unroll
    for (int p = 0; p < NP_R; ++p) {      // input part of the V_r weights into the same buffer
        const int k = kr + 16 * p;
        if (k < IN && (tid & 31) < Dq) st4(sWr + k * FF_LDR + 4 * nq, wr1[p]);
    }
    if (aq < INq) {
        if (l == 0 && m.drop_e > 0.f) {
            const float4 mk = drop_mult4(m.seed, (unsigned)g, G4R_STREAM_DROP_EMBED, m0 + ar, aq, 1.0f - m.drop_e);
            ay.x *= mk.x; ay.y *= mk.y; ay.z *= mk.z; ay.w *= mk.w;
        }
        if (!arow_ok) ay = zero4;
        float2* d = reinterpret_cast<float2*>(sA + ar * LDA + 4 * aq);
        d[0] = make_float2(ay.x, ay.y); d[1] = make_float2(ay.z, ay.w);
        // the dense-gradient tiles read the (dropout-masked) layer-0 input rows back (dWx = yin^T dV)
        if (l == 0 && blockIdx.x == 0 && arow_ok) st4(m.yin0 + (size_t)(m0 + ar) * IN + 4 * aq, ay);
    }
    __syncthreads();
    if (clk && tid == 0) clk[6] = wall_clock64();
    if (wid < NT1) accR = mma(accR, paA, sWr + lg * FF_LDR + wid * 16 + li, FF_LDR, IN);
    if (doZ) accT = mma(accT, paA, sWz + lg * FF_LDT + tsub * 16 + li, FF_LDT, IN);
    if (doC) accT = mma(accT, paA, sWc + lg * FF_LDT + tsub * 16 + li, FF_LDT, IN);
    if (clk && tid == 0) clk[7] = wall_clock64();
    // epilogue A
    GAS float *rb = m.r[l], *Hrb = m.Hr[l], *zb = m.z[l];
    if (wid < NT1) {
        const bool mine = (nr >= n0 && nr < n0 + 32);      // this column tile stores its own 32 columns of r / Hr
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int r = 4 * lg + rg, row = m0 + r;
            if (nr < D) {
                const float rr = sigmoidf_(accR[rg] + b_r), hr = sA[r * LDA + IN + nr] * rr;
                sHr[r * LDH + nr] = hr;
                if (mine && row < M) { rb[(size_t)row * D + nr] = rr; Hrb[(size_t)row * D + nr] = hr; }
            }
        }
    }
    if (doZ || doC) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int r = 4 * lg + rg, row = m0 + r;
            if (doZ) {
                const float zz = sigmoidf_(accT[rg] + b_z);
                sZ[r * 33 + tsub * 16 + li] = zz;
                if (nt < D && row < M) zb[(size_t)row * D + nt] = zz;
            } else {
                sVc[r * 33 + tsub * 16 + li] = accT[rg] + b_c;
            }
        }
    }
    __syncthreads();
    if (clk && tid == 0) clk[8] = wall_clock64();
    // ---- stage B: (H r) Wh for the tile's columns; wave w: sub-tile (w & 1), quarter (w >> 1) of K = D
    const int kq = wid >> 1;
    const int kquart = ((Dq + 3) >> 2) << 2, kb = kq * kquart, ke = min(D, kb + kquart);
    f32x4 accB = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kb < ke) accB = mma(accB, sHr + li * LDH + lg + kb, sWh + (kb + lg) * FF_LDT + tsub * 16 + li, FF_LDT, ke - kb);
    if (clk && tid == 0) clk[9] = wall_clock64();
    if (kq) sJ[(wid - 2) * 64 + lane] = accB;
    __syncthreads();
    if (kq) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) {      // quarters 1..3 in order
        const f32x4 o = sJ[(2 * j + tsub) * 64 + lane];
        accB[0] += o[0]; accB[1] += o[1]; accB[2] += o[2]; accB[3] += o[3];
    }
    if (nt >= D) return;
    GAS float *cl = m.c[l], *hout = m.hd[l], *Hnext = m.H[l][(g + 1) & 1];
    const float drop_h = m.drop_h;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * lg + rg, row = m0 + r;
        if (row >= M) continue;
        const size_t o = (size_t)row * D + nt;
        const float cc = act_fwd(m.hidden_act, m.ha_p0, m.ha_p1, accB[rg] + sVc[r * 33 + tsub * 16 + li]);
        const float zz = sZ[r * 33 + tsub * 16 + li], hprev = sA[r * LDA + IN + nt];
        float h = (1.0f - zz) * hprev + zz * cc;
        if (drop_h > 0.f) h *= drop_mult(m.seed, (unsigned)g, G4R_STREAM_DROP_HIDDEN + l, row, nt, 1.0f - drop_h);
        cl[o] = cc;
        hout[o] = h;
        Hnext[o] = ((rst4 >> (8 * rg)) & 0xFF) ? 0.f : h;
    }
    if (clk && tid == 0) clk[10] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// Scoring GEMM: Sc[B, N] = h[B, D] * Wy[items]^T + By[items] - logq * lq[items]    (gru4rec.py:493-495)
// 64 x 32 tiles; the B provider gathers the TN output-embedding rows of the tile's columns (in-batch targets,
// then the step's row of the negative-sample store).  Publishes the column -> item map for the later kernels.
#ifndef SF_BM
#define SF_BM 64
#endif
#ifndef T2_BK
#define T2_BK 16     // K chunk of the gemm_tile2 kernels
#endif
#ifndef T3_NST
#define T3_NST 3     // ring depth of the gemm_tile3 (LDS-DMA) kernels
#endif
#ifndef T3_BKS
#define T3_BKS 32    // k per ring stage (measured at B = 512, N = 8704, D = 256 / B = 240, N = 2288, D = 512, us: 3 x 32: 31.3 / 15.0,
                     // 4 x 16: 32.3 / 16.1, 5 x 16: 32.7 / 16.4 -- gemm_tile2: 34.1 / 17.5)
#endif

// T2 > 3: the gemm_tile2 variant (64 x 64 tiles, mfma 32x32x2) with K chunks of T2 floats; T2 == 3: gemm_tile3 (LDS-DMA ring);
// TBN / TBK then only name the instance
template <int TBN, int TBK, int T2 = 0>
__global__ __launch_bounds__(GT_NTH) void k_score_fwd(const DevModel* __restrict__ mp, StepState* st) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    constexpr int SMEM_TILE = T2 == 3 ? Tile3Cfg<T3_NST, T3_BKS>::SMEM_FLOATS : T2 ? Tile2Cfg<(T2 > 3) ? T2 : 16>::SMEM_FLOATS : TileCfg<SF_BM, TBN, TBK, false, true>::SMEM_FLOATS;
    int* sItem = reinterpret_cast<int*>(smem + SMEM_TILE);   // [TBN]
    const int tid = threadIdx.x;
    // in-kernel phase trace (tools/clk_score.py), gemm_tile2 variant only: in the small-shape variant the test of the descriptor
    // field in front of everything else cost 0.5 us per launch
    GAS long long* trc = nullptr;
    if constexpr (T2 != 0) {
        const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
        trc = (G4R_DBGTILE(m) && wgid < 2048) ? G4R_DBGTILE(m) + 8 * (size_t)(4096 + wgid) : nullptr;
        if (trc && tid == 0) trc[0] = wall_clock64();
    }
    const StepCtx c = load_ctx(st);
    const int M = c.M, B = m.B, D = m.Dtop, N = m.N;
    // the row tiles of a column tile run on ONE XCD (they share the gathered Wy rows of the tile's columns): tile order = column
    // tile major within an XCD's contiguous range (g4r_device.cuh: xcd_tile)
    // (the 64 x 64 variants keep the plain order: their launches have a multiple of 8 column tiles per row of tiles, which already
    // puts a column tile's row tiles on one XCD, and the row-major start order measured 0.7 us better at B = 512, N = 8704)
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (T2 == 0) {
        const int tile = G4R_XCD_TILE(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        bx = tile / (int)gridDim.y; by = tile - bx * (int)gridDim.y;
    }
    const int n0 = bx * TBN, m0 = by * SF_BM;
    if (tid < TBN) {
        const int n = n0 + tid;
        int item = m.cur_col[min(n, m.ldSc - 1)];      // targets | samples of this step, staged by the previous step's bookkeeping
        if (n >= m.ldSc) item = -1;
        if constexpr (T2 == 0) sItem[tid] = item;
        if (by == 0 && n < m.ldSc) {
            m.col_item[n] = item;
            if (n < N) {
                m.occ_idx[B + n] = item;
                if (item >= 0 && m.xmode == 0) {
                    int* fl = (int*)m.occ_fl + 4 * (size_t)item;
                    atomicMax(fl, B + n + 1);
                    atomicMax(fl + 1, m.R - (B + n));
                    atomicAdd(fl + 2, 1);
                }
            }
        }
    }
    if (m0 >= M) return;
    __syncthreads();
    if (trc && tid == 0) { trc[1] = wall_clock64(); trc[5] = c.t; }      // step context + column items here
    const GAS float* hsrc = m.hd[m.n_layers - 1];
    const GAS float *Wy = m.Wy, *By = m.By, *lq_tgt = m.lq_tgt, *lq_smp = m.lq_smp;
    GAS float* Sc = m.Sc;
    const float logq = m.logq;
    const int ldSc = m.ldSc;
    auto aload = [&](int kk, int r, int cc) -> float4 {
        const int row = m0 + r, k = kk + cc;
        return ld4_if(hsrc, (size_t)row * D + k, row < M && k < D);
    };
    auto bload = [&](int kk, int r, int cc) -> float4 {
        const int item = sItem[r], k = kk + cc;
        return ld4_if(Wy, (size_t)max(item, 0) * D + k, item >= 0 && k < D);
    };
    auto pre = [&](int row, int n) -> float4 {      // bias - logQ correction of the column's item
        const int item = (n < N) ? sItem[n - n0] : -1;
        const bool ok = item >= 0;
        float x = ldf_at(By, max(item, 0), ok);
        const bool lq = ok && logq != 0.f;      // branch-free: the logQ table is only touched when it exists
        x -= logq * ldf_at(lq ? (n < B ? lq_tgt : lq_smp) : By, max(item, 0), lq);
        return make_float4(x, 0.f, 0.f, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M || n >= N) return;
        Sc[(size_t)row * ldSc + n] = v + p.x;
    };
    if constexpr (T2 != 0) {      // long score rows / big batches, D a multiple of T2 (host)
        // the LDS tile fills the workgroup's 32 KiB: column items come straight from the staged list (L2), not from sItem
        const GAS int* ccol = m.cur_col;
        const int ldc = m.ldSc;
        auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? hsrc + (size_t)(m0 + r) * D : nullptr; };
        auto brow = [&](int r) -> const GAS float* {
            const int item = (n0 + r < ldc) ? ccol[n0 + r] : -1;
            return item >= 0 ? Wy + (size_t)item * D : nullptr;
        };
        auto pre2 = [&](int row, int n) -> float4 {
            const int item = (n < N) ? ccol[min(n, ldc - 1)] : -1;
            const bool ok = item >= 0;
            float x = ldf_at(By, max(item, 0), ok);
            const bool lq = ok && logq != 0.f;
            x -= logq * ldf_at(lq ? (n < B ? lq_tgt : lq_smp) : By, max(item, 0), lq);
            return make_float4(x, 0.f, 0.f, 0.f);
        };
        if constexpr (T2 == 3) gemm_tile3<T3_NST, T3_BKS, true>(m0, n0, D, arow, brow, m.zrow, pre2, epi, smem, trc);
        else gemm_tile2<(T2 > 3) ? T2 : 16, true>(m0, n0, D, arow, brow, m.zrow, pre2, epi, smem, trc);
    } else gemm_tile<SF_BM, TBN, TBK, false, true, GT_NTH>(m0, n0, D, aload, bload, pre, epi, smem);
}

// ---------------------------------------------------------------------------------------------
// Per-row final activation, loss and d cost / d s, in place in Sc.  One 1024-thread workgroup per batch row; the row's
// yhat and softmax numerators live in LDS (every thread only revisits the columns it wrote itself, so the passes need
// no barriers besides the three block reductions); row statistics via DPP wave reductions.
// Column j is active iff j < M (in-batch targets) or j >= B (sampled negatives); row i's positive is
// column i.  Losses: gru4rec.py:225-230 (cross_entropy), :239-241 (bpr_max), :245-248 (top1_max),
// softmax_neg :199-203.  The gradient goes through the softmax weights, as T.grad does.
#ifndef LOSS_T
#define LOSS_T 1024
#endif
#define LOSS_NW (LOSS_T / 64)
// NV simultaneous block sums / maxima; `red` = NV * LOSS_NW floats that no other reduction of the kernel touches
template <int NV, bool MAX>
__device__ __forceinline__ void block_reduce(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = MAX ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[q * LOSS_NW + w] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float a = red[q * LOSS_NW];
#pragma unroll
        for (int u = 1; u < LOSS_NW; ++u) a = MAX ? fmaxf(a, red[q * LOSS_NW + u]) : a + red[q * LOSS_NW + u];
        v[q] = a;
    }
}

__device__ __forceinline__ float softplusf_(float x) {      // log(1 + e^x), stable
    return fmaxf(x, 0.f) + log1pf(fexp(-fabsf(x)));
}

// LONG_ROW (score rows whose two copies do not fit the LDS, > ~19 K columns): `se` lives in the row's own memory instead -- a thread
// has read its columns' scores before it writes anything there, and only ever revisits its own columns.
// SPEC: the (final activation, loss) pair as a compile-time constant for the pairs BASELINE's configurations use -- 1 elu + bpr-max,
// 2 softmax + cross-entropy, 3 elu + top1-max; 0 = any pair, read from the descriptor.  The element loops below switch on both for
// every element (eight scalar branches per element and pass in the generic build); with constants the switches fold away.
// V floats of a row at once (V = 1 or 4: one 16-byte global / LDS access per four columns)
template <int V, class Ptr>
__device__ __forceinline__ void vld(float (&o)[V], Ptr p) {
    if constexpr (V == 4) { const float4 t = ld4(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
    else o[0] = p[0];
}
template <int V, class Ptr>
__device__ __forceinline__ void vst(Ptr p, const float (&o)[V]) {
    if constexpr (V == 4) st4(p, make_float4(o[0], o[1], o[2], o[3]));
    else p[0] = o[0];
}

// V: columns per thread and loop trip.  V = 1: thread t takes columns t, t + 1024, ... (short rows: every thread has a column);
// V = 4: columns 4 t .. 4 t + 3, then + 4096 (long rows: the element loops are VALU-issue bound there -- 512 rows x 8704 columns cost
// ~107 instructions per element in the one-column form, loop control, address arithmetic and predication around 4-byte accesses;
// the four-column form shares them between four elements).  Every pass works on ALL columns of its groups: inactive ones compute on
// a harmless stand-in and are masked where they would enter a sum, a maximum or the row in memory (selects, no branches) -- and a
// group of V columns that lies wholly inside the active targets or the negatives and does not hold the row's positive (all but a
// handful per row) takes a copy of the loop body compiled WITHOUT those selects (`fast`).
template <bool LONG_ROW, int SPEC, int V>
__global__ __launch_bounds__(LOSS_T) void k_loss_rows(const DevModel* __restrict__ mp, StepState* st) {
    const DevModel& m = *mp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int B = m.B, N = m.N, i = blockIdx.x;
    const int fact = SPEC == 1 || SPEC == 3 ? (int)G4R_ACT_ELU : (SPEC == 2 ? (int)G4R_ACT_SOFTMAX : m.final_act);
    const int lossk = SPEC == 1 ? (int)G4R_LOSS_BPR_MAX : (SPEC == 2 ? (int)G4R_LOSS_XE : (SPEC == 3 ? (int)G4R_LOSS_TOP1_MAX : m.loss));
    const int ldSc = m.ldSc;
    const float fp0 = m.fa_p0, fp1 = m.fa_p1, invB = m.inv_B, bpreg = m.bpreg, smooth = m.smoothing;
    GAS float* row = m.Sc + (size_t)i * ldSc;
    float* sy = smem;                  // [ldSc] yhat
    std::conditional_t<LONG_ROW, GAS float*, float*> se;      // [ldSc] softmax numerators, later d L / d yhat
    if constexpr (LONG_ROW) se = row; else se = smem + ldSc;
    float* red = smem + (LONG_ROW ? 1 : 2) * ldSc;      // [8][3 * LOSS_NW] one region per reduction
    constexpr int STEP = V * LOSS_T;
    const int jt = V * tid;            // this thread's first column; its columns are jt + k STEP + (0 .. V - 1)
    // The first LOSS_PRE groups of every thread are requested TOGETHER with the step state (row i exists for every i < B), so
    // the kernel starts with one memory round trip instead of two (state -> M -> predicated row loads); M only masks them.
    // (measured, round 3: requesting the WHOLE row up front -- 10 scores per thread at B = 512 with 8192 negatives -- does not move
    // the kernel, 17.0 vs 17.1 us; neither do 512- or 256-thread workgroups, 18.7 / 29.5 us: the row is not waiting for its loads)
    constexpr int LOSS_PRE = (V == 4) ? 2 : 4;
    const StepCtx c = load_ctx(st);
    float pre_s[LOSS_PRE][V];
#pragma unroll
    for (int q = 0; q < LOSS_PRE; ++q) vld<V>(pre_s[q], row + min(jt + q * STEP, ldSc - V));
    const int M = c.M;
    if (i >= M) return;
    const bool fsm = (fact == G4R_ACT_SOFTMAX), fsl = (fact == G4R_ACT_SOFTMAX_LOGIT);
    const float n_out = (float)(M + (N - B));      // active columns (gru4rec.py:227,233,244: M + n_sample)
    // Column j is active iff j < M (in-batch targets) or B <= j < N (sampled negatives)
    auto active = [&](int j) { return j < N && (j < M || j >= B); };
    auto grp_fast = [&](int j0) { return (j0 >= B && j0 + V <= N) || (j0 + V <= M && (i < j0 || i >= j0 + V)); };
    using Fast = std::true_type;
    using Slow = std::false_type;
#define G4R_GROUPS(lim, fn) for (int j0 = jt; j0 < (lim); j0 += STEP) { if (grp_fast(j0)) fn(Fast{}, j0); else fn(Slow{}, j0); }
    // ---- final activation (gru4rec.py:496); mneg = max over the negatives of yhat (with the positive as a 0)
    float mneg[1] = {0.f};
    if (fsm || fsl) {
        float mx[1] = {-INFINITY};
        auto first = [&](auto F, int j0, const float (&v)[V]) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const bool a = F.value || active(j0 + e);
                o[e] = a ? v[e] : 0.f;
                mx[0] = a ? fmaxf(mx[0], v[e]) : mx[0];
            }
            vst<V>(sy + j0, o);
        };
#pragma unroll
        for (int q = 0; q < LOSS_PRE; ++q) {
            const int j0 = jt + q * STEP;
            if (j0 < N) { if (grp_fast(j0)) first(Fast{}, j0, pre_s[q]); else first(Slow{}, j0, pre_s[q]); }
        }
        for (int j0 = jt + LOSS_PRE * STEP; j0 < N; j0 += STEP) {
            float v[V];
            vld<V>(v, row + j0);
            if (grp_fast(j0)) first(Fast{}, j0, v); else first(Slow{}, j0, v);
        }
        block_reduce<1, true>(mx, red);
        float sm[1] = {0.f};
        auto numer = [&](auto F, int j0) {
            float y[V];
            vld<V>(y, sy + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const bool a = F.value || active(j0 + e);
                const float ex = fexp(a ? y[e] - mx[0] : 0.f);
                if (fsm) y[e] = ex;
                sm[0] += a ? ex : 0.f;
            }
            if (fsm) vst<V>(sy + j0, y);
        };
        G4R_GROUPS(N, numer)
        block_reduce<1, false>(sm, red + 3 * LOSS_NW);
        const float inv_z = 1.f / sm[0], lse = logf(sm[0]);
        auto norm = [&](auto F, int j0) {
            float y[V];
            vld<V>(y, sy + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int j = j0 + e;
                // softmax :193-195 ; softmax_logit :196-198 = log(sum exp(x - max)) - (x - max)
                const float yy = fsm ? y[e] * inv_z : lse - (y[e] - mx[0]);
                y[e] = yy;
                mneg[0] = (F.value || (active(j) && j != i)) ? fmaxf(mneg[0], yy) : mneg[0];
            }
            vst<V>(sy + j0, y);
        };
        G4R_GROUPS(N, norm)
    } else {
        auto first = [&](auto F, int j0, const float (&v)[V]) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int j = j0 + e;
                const bool a = F.value || active(j);
                const float y = act_fwd_sel(fact, fp0, fp1, a ? v[e] : 0.f);
                o[e] = y;
                mneg[0] = (F.value || (a && j != i)) ? fmaxf(mneg[0], y) : mneg[0];
            }
            vst<V>(sy + j0, o);
        };
#pragma unroll
        for (int q = 0; q < LOSS_PRE; ++q) {
            const int j0 = jt + q * STEP;
            if (j0 < N) { if (grp_fast(j0)) first(Fast{}, j0, pre_s[q]); else first(Slow{}, j0, pre_s[q]); }
        }
        for (int j0 = jt + LOSS_PRE * STEP; j0 < N; j0 += STEP) {
            float v[V];
            vld<V>(v, row + j0);
            if (grp_fast(j0)) first(Fast{}, j0, v); else first(Slow{}, j0, v);
        }
    }
    block_reduce<1, true>(mneg, red + 6 * LOSS_NW);      // its barrier also publishes sy[i]
    const float yd = sy[i];
    const bool own_i = tid == ((i / V) % LOSS_T);       // the thread whose columns include i
    float Lrow = 0.f;
    // ---- loss and d L / d yhat_j -> se[j] (every thread its own columns)
    if (lossk == G4R_LOSS_XE && fsm && smooth == 0.f) {
        // fused softmax + cross-entropy: ds_k = yhat_k * (dy_k - sum_j dy_j yhat_j) with dy = -delta_ik / (yd + eps)
        Lrow = -logf(yd + G4R_EPS_LOSS);
        const float coef = yd / (yd + G4R_EPS_LOSS);
        auto grad = [&](auto F, int j0) {
            float y[V], o[V];
            vld<V>(y, sy + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int j = j0 + e;
                if (F.value) o[e] = coef * y[e] * invB;
                else o[e] = active(j) ? coef * (y[e] - (j == i ? 1.f : 0.f)) * invB : 0.f;
            }
            vst<V>(row + j0, o);
        };
        G4R_GROUPS(ldSc, grad)
        if (tid == 0) m.lossrow[i] = Lrow;
        return;
    }
    if (lossk == G4R_LOSS_XE || lossk == G4R_LOSS_XE_LOGIT) {
        // cross_entropy :225-230 on probabilities, cross_entropy_logits :231-236 on -log-probabilities, with label
        // smoothing: (1 - n/(n-1) s) * l(yd) + s/(n-1) * sum_j l(y_j)
        const bool lg = (lossk == G4R_LOSS_XE_LOGIT);
        const float wd = 1.f - n_out / (n_out - 1.f) * smooth, wa = smooth / (n_out - 1.f);
        float sa[1] = {0.f};
        for (int j0 = jt; j0 < N; j0 += STEP) {
            float y[V], d[V];
            vld<V>(y, sy + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int j = j0 + e;
                const bool a = active(j);
                const float yy = a ? y[e] : 1.f;
                float dd = 0.f;
                if (smooth != 0.f) { sa[0] += a ? (lg ? yy : -logf(yy + G4R_EPS_LOSS)) : 0.f; dd = lg ? wa : -wa / (yy + G4R_EPS_LOSS); }
                if (j == i) dd += lg ? wd : -wd / (yy + G4R_EPS_LOSS);
                d[e] = dd;
            }
            vst<V>(se + j0, d);
        }
        if (smooth != 0.f) block_reduce<1, false>(sa, red + 9 * LOSS_NW);
        Lrow = wd * (lg ? yd : -logf(yd + G4R_EPS_LOSS)) + wa * sa[0];
    } else if (lossk == G4R_LOSS_BPR || lossk == G4R_LOSS_TOP1) {
        float s[2] = {0.f, 0.f};
        if (lossk == G4R_LOSS_BPR) {
            // bpr :237-238: sum over ALL active columns of -log sigmoid(yd - y_j) (the diagonal adds log 2)
            for (int j0 = jt; j0 < N; j0 += STEP) {
                float y[V], d[V];
                vld<V>(y, sy + j0);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int j = j0 + e;
                    const bool a = active(j);
                    const float yy = a ? y[e] : yd;
                    s[0] += a ? softplusf_(yy - yd) : 0.f;
                    const float dd = (a && j != i) ? sigmoidf_(yy - yd) : 0.f;
                    s[1] += dd;
                    d[e] = dd;
                }
                vst<V>(se + j0, d);
            }
            block_reduce<2, false>(s, red + 9 * LOSS_NW);
            Lrow = s[0];
            if (own_i) se[i] = -s[1];
        } else {
            // top1 :242-244: mean_j (sigmoid(y_j - yd) + sigmoid(y_j^2)) - sigmoid(yd^2) / n  (the diagonal leaves 0.5 / n).
            // As written in the reference the (M,) mean minus the (M, 1) diagonal term broadcasts to (M, M) before the
            // sum, i.e. the cost is M times the per-row formula; reproduced here (wM).
            const float inv_n = 1.f / n_out, wM = (float)M;
            for (int j0 = jt; j0 < N; j0 += STEP) {
                float y[V], d[V];
                vld<V>(y, sy + j0);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int j = j0 + e;
                    const bool a = active(j) && j != i;
                    const float yy = a ? y[e] : 0.f;
                    const float u = sigmoidf_(yy - yd), q = sigmoidf_(yy * yy);
                    s[0] += a ? u + q : 0.f;
                    s[1] += a ? u * (1.f - u) : 0.f;
                    d[e] = a ? wM * inv_n * (u * (1.f - u) + 2.f * yy * q * (1.f - q)) : 0.f;
                }
                vst<V>(se + j0, d);
            }
            block_reduce<2, false>(s, red + 9 * LOSS_NW);
            Lrow = wM * inv_n * (s[0] + 0.5f);
            if (own_i) se[i] = -wM * inv_n * s[1];
        }
    } else {
        // softmax over the negatives, with the positive zeroed first (so the max includes a 0)
        const float mx = mneg[0];
        // sigmoid(yd - y_j) = 1 / (1 + exp(y_j - yd)) = 1 / (1 + e_j c) with the softmax numerator e_j = exp(y_j - mx) and the row
        // constant c = exp(mx - yd): one exp per element serves both.  c is clamped so that an underflowed e_j = 0 gives 0 * c = 0
        // (sigma = 1, and p_j = 0 anyway).  The row statistics A = sum sigma p, Q = sum y^2 p, ... are linear in p = e / Z, so
        // their unnormalised sums are taken in the SAME pass as Z = sum e and divided afterwards: one pass over the row and one
        // block reduction less than "Z first, then the statistics".
        const float cexp = fexp(fminf(mx - yd, 80.f));
        float s[4] = {0.f, 0.f, 0.f, 0.f};      // Z, and unnormalised A / T, Q, sum sigma' e
        auto stats = [&](auto F, int j0) {
            float y[V], ev[V];
            vld<V>(y, sy + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int j = j0 + e;
                const bool a = F.value || (active(j) && j != i);
                const float yy = a ? y[e] : mx;
                const float ex = a ? fexp(yy - mx) : 0.f;      // (an inactive column: e = 0 leaves every sum alone)
                ev[e] = ex;
                s[0] += ex;
                if (lossk == G4R_LOSS_BPR_MAX) {
                    const float sg = frcp(1.0f + ex * cexp);
                    s[1] += sg * ex;                 // A Z
                    s[2] += yy * yy * ex;            // Q Z
                    s[3] += sg * (1.f - sg) * ex;    // (sum sigma' p) Z
                } else {
                    const float u = 1.0f - frcp(1.0f + ex * cexp), q = sigmoidf_(yy * yy);
                    s[1] += ex * (u + q);            // T Z
                    s[3] += ex * u * (1.f - u);
                }
            }
            vst<V>(se + j0, ev);
        };
        G4R_GROUPS(N, stats)
        block_reduce<4, false>(s, red + 9 * LOSS_NW);
        const float inv_sm = 1.f / s[0];
        s[0] = s[1] * inv_sm; s[1] = s[2] * inv_sm; s[2] = s[3] * inv_sm;
        const float s1 = s[0], s2 = s[1], s3 = s[2];
        float dyd;
        const float inv_A = 1.f / (s1 + G4R_EPS_LOSS);
        if (lossk == G4R_LOSS_BPR_MAX) {
            Lrow = -logf(s1 + G4R_EPS_LOSS) + bpreg * s2;
            dyd = -s3 * inv_A;
        } else {
            Lrow = s1;
            dyd = -s3;
        }
        auto dLn = [&](float y, float ex) -> float {      // d L / d yhat_j of a NEGATIVE from its yhat and softmax numerator
            const float p = ex * inv_sm;
            if (lossk == G4R_LOSS_BPR_MAX) {
                const float sg = frcp(1.0f + ex * cexp);
                return -p * (sg - sg * (1.f - sg) - s1) * inv_A + bpreg * p * (2.f * y + y * y - s2);
            }
            const float u = 1.0f - frcp(1.0f + ex * cexp), q = sigmoidf_(y * y);
            return p * (u + q - s1) + p * (u * (1.f - u) + 2.f * y * q * (1.f - q));
        };
        auto dL = [&](int j, float y, float ex) -> float { const float d = dLn(y, ex); return j == i ? dyd : d; };
        if (!(fsm || fsl)) {
            // element-wise final activation (the usual partner of these losses): d cost / d s = dL f'(s) needs no row sum, so the
            // gradient goes straight to the row in memory -- one pass over the row (a store and a load of se per element) less
            auto grad = [&](auto F, int j0) {
                float y[V], ev[V], o[V];
                vld<V>(y, sy + j0);
                vld<V>(ev, se + j0);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int j = j0 + e;
                    if (F.value) o[e] = dLn(y[e], ev[e]) * act_bwd_from_out(fact, fp0, fp1, y[e]) * invB;
                    else {
                        const bool a = active(j);
                        const float yy = a ? y[e] : 0.f, ex = a ? ev[e] : 0.f;
                        o[e] = a ? dL(j, yy, ex) * act_bwd_from_out(fact, fp0, fp1, yy) * invB : 0.f;
                    }
                }
                vst<V>(row + j0, o);
            };
            G4R_GROUPS(ldSc, grad)
            if (tid == 0) m.lossrow[i] = Lrow;
            return;
        }
        for (int j0 = jt; j0 < N; j0 += STEP) {
            float y[V], ev[V];
            vld<V>(y, sy + j0);
            vld<V>(ev, se + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const bool a = active(j0 + e);
                ev[e] = dL(j0 + e, a ? y[e] : 0.f, a ? ev[e] : 0.f);
            }
            vst<V>(se + j0, ev);
        }
    }
    // ---- d L / d yhat -> d cost / d s through the final activation, straight to the row in memory (inactive and
    // padding columns get 0).  softmax: y (d - sum_j d_j y_j); softmax_logit: softmax_k sum_j d_j - d_k with
    // softmax_k = exp(-yhat_k); element-wise: d f'(s)
    float inner[1] = {0.f};
    if (fsm || fsl) {
        for (int j0 = jt; j0 < N; j0 += STEP) {
            float y[V], d[V];
            vld<V>(y, sy + j0);
            vld<V>(d, se + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) inner[0] += active(j0 + e) ? (fsm ? d[e] * y[e] : d[e]) : 0.f;
        }
        block_reduce<1, false>(inner, red + 15 * LOSS_NW);
    }
    auto grad = [&](auto F, int j0) {
        float y[V], d[V], o[V];
        vld<V>(y, sy + j0);
        vld<V>(d, se + j0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const bool a = F.value || active(j0 + e);
            const float yy = a ? y[e] : 0.f, dd = a ? d[e] : 0.f;
            float out;
            if (fsm) out = yy * (dd - inner[0]);
            else if (fsl) out = fexp(-yy) * inner[0] - dd;
            else out = dd * act_bwd_from_out(fact, fp0, fp1, yy);
            o[e] = a ? out * invB : 0.f;
        }
        vst<V>(row + j0, o);
    };
    G4R_GROUPS(ldSc, grad)
    if (tid == 0) m.lossrow[i] = Lrow;
#undef G4R_GROUPS
}

// ---------------------------------------------------------------------------------------------
// Scoring backward, two roles in one launch (block ranges):
//   role A (blockIdx.x < nblkA): dSy[N, D] = ds^T h over 32x32 tiles (A = ds read as [k = b][m = n]); the spare
//          column d == D of the last d-tile carries a ones column of h, so it accumulates dSBy = colsum(ds).
//   role B: split-K slabs of dh = ds * Sy: tile (32 rows b, 32 cols d) x one 128-wide chunk of score columns,
//          B provider = gathered Wy rows of the chunk's columns.  Slabs are summed (fixed order) by k_gru_bwd_pre.
// TB x TB output tiles, TBK-deep K chunks: 32 / 128 for the RSC15-sized step (more workgroups than CUs matter there),
// 64 / 64 for long score rows and big batches (twice the flops per operand byte pulled from L2).
template <int TB, int TBK>
__global__ __launch_bounds__(GT_NTH) void k_score_bwd(const DevModel* __restrict__ mp, StepState* st, int nblkA, int ndtA, int ndtB, int nrtB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int M = c.M, B = m.B, D = m.Dtop, N = m.N, ld = m.ldSc, tid = threadIdx.x;
    const GAS float* h = m.hd[m.n_layers - 1];
    const GAS float* Sc = m.Sc;
    const GAS float* Wy = m.Wy;
    // column -> item map of the tile's score columns, staged in LDS (the gathers must not chain behind index loads)
    int* sIt = reinterpret_cast<int*>(smem + max(TileCfg<TB, TB, TBK, true, false>::SMEM_FLOATS,
                                                  TileCfg<TB, TB, TBK, false, false>::SMEM_FLOATS));
    if ((int)blockIdx.x < nblkA) {
        const int tile = G4R_XCD_TILE(blockIdx.x, nblkA);
        const int nt = tile / ndtA, dt = tile - nt * ndtA;
        const int n0 = nt * TB, d0 = dt * TB;
        if (tid < TB) sIt[tid] = (n0 + tid < N) ? m.col_item[n0 + tid] : -1;
        __syncthreads();
        auto aload = [&](int kk, int r, int cc) -> float4 {      // staging tile [k = b][m = n]
            const int b = kk + r, n = n0 + cc;
            return ld4_if(Sc, (size_t)b * ld + n, b < M && n < ld);
        };
        auto bload = [&](int kk, int r, int cc) -> float4 {      // [k = b][n = d], ones in column d == D
            const int b = kk + r, d = d0 + cc;
            float4 v = ld4_if(h, (size_t)b * D + d, b < M && d < D);
            if (b < M && d == D) v.x = 1.f;
            return v;
        };
        // the per-occurrence Adagrad scaling (gru4rec.py:335-340) happens here, in parallel over all occurrences:
        // every occurrence uses the PRE-step accumulator of its item, so the steps are independent; the sparse
        // kernel only has to add them up in occurrence order.
        // An item that occurs ONCE among the step's gathered rows (count field of its occ_fl entry, complete since the forward
        // kernels; ~80 % of the occurrences) gets its new accumulator written IN PLACE right here -- this epilogue holds acc[item]
        // already -- so that the update kernel moves three rows for it (step read, parameter read + write) instead of five, and the
        // dA plane is only written for items with several occurrences (all of which must see the PRE-step accumulator: they go
        // through dA and the owner wave of the update kernel as before).  The parameter itself cannot be written here: role B of
        // this launch gathers the same Wy rows.
        GAS float *accWy = m.accWy, *accBy = m.accBy;
        const GAS int* occ_fl = m.occ_fl;
        GAS float *dSy = G4R_DSY(m, c.g), *dAy = m.dAy, *dSBy = G4R_DSBY(m, c.g), *dABy = m.dABy;
        const float lr = m.lr;
        const bool generic = m.generic != 0;
        auto pre = [&](int n, int d) -> float4 {
            const int item = (n - n0 < TB) ? sIt[n - n0] : -1;
            const bool ok = item >= 0 && d <= D;
            const float a = (d < D) ? ldf_at(accWy, (size_t)max(item, 0) * D + d, ok) : ldf_at(accBy, max(item, 0), ok);
            const int cnt = occ_fl[4 * (size_t)max(item, 0) + 2];
            return make_float4(a, ok ? 1.f : 0.f, __int_as_float(cnt), 0.f);
        };
        auto epi = [&](int n, int d, float g, float4 p) {
            if (n >= N || d > D) return;
            const float an = p.x + G4R_MUT_ACC(g * g);
            float step = (p.y != 0.f) ? G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD)) : 0.f;
            if (generic) step = (p.y != 0.f) ? g : 0.f;      // raw per-occurrence gradient: the update kernel applies the rule
            const bool single = !generic && p.y != 0.f && __float_as_int(p.z) == 1;
            const int item = single ? sIt[n - n0] : 0;
            if (d < D) {
                dSy[(size_t)n * D + d] = step;
                if (single) accWy[(size_t)item * D + d] = an; else dAy[(size_t)n * D + d] = an;
            } else {
                dSBy[n] = step;
                if (single) accBy[item] = an; else dABy[n] = an;
            }
        };
        gemm_tile<TB, TB, TBK, true, false, GT_NTH>(n0, d0, M, aload, bload, pre, epi, smem);
        return;
    }
    const int w = G4R_XCD_TILE(blockIdx.x - nblkA, (int)gridDim.x - nblkA);
    const int per_kc = nrtB * ndtB;
    const int kc = w / per_kc, rem = w - kc * per_kc, rt = rem / ndtB, dt = rem - rt * ndtB;
    // a slab covers kch = (multiple of TBK) score columns: long score rows (many negatives) use wider slabs so that the
    // number of split-K partials, and the traffic of writing and re-reading them, stays ~17 (host: d.kch)
    const int kch = m.kch;
    const int m0 = rt * TB, d0 = dt * TB, kbeg = kc * kch;
    if (m0 >= M) return;
    for (int i = tid; i < kch; i += (int)blockDim.x) sIt[i] = (kbeg + i < ld) ? m.col_item[kbeg + i] : -1;
    __syncthreads();
    GAS float* dhpart = m.dhpart;
    auto aload = [&](int kk, int r, int cc) -> float4 {
        const int b = m0 + r, n = kbeg + kk + cc;
        return ld4_if(Sc, (size_t)b * ld + n, b < M && n < ld);
    };
    auto bload = [&](int kk, int r, int cc) -> float4 {
        const int item = sIt[kk + r], d = d0 + cc;
        return ld4_if(Wy, (size_t)max(item, 0) * D + d, item >= 0 && d < D);
    };
    auto epi = [&](int b, int d, float v, float4) {
        if (b < M && d < D) dhpart[((size_t)kc * B + b) * D + d] = v;
    };
    gemm_tile<TB, TB, TBK, false, false, GT_NTH>(m0, d0, min(kch, ld - kbeg), aload, bload, NoPre(), epi, smem);
}

// Scoring backward for long score rows / big batches (B >= 256, >= 4096 score columns, D a multiple of 64) on gemm_tile2k: 64 x 64
// tiles, v_mfma_f32_32x32x2_f32, 16-deep double-buffered chunks.  Three roles in one launch (block ranges):
//   A  [0, nblkA)               dSy[n0.., d0..] = ds^T h over the batch (both operands K-major); epilogue as k_score_bwd role A
//   B  [nblkA, nblkA + nblkB)   split-K slab kc of dh = ds Sy (ds K-contiguous, gathered Wy rows K-major)
//   C  the rest                 dSBy = column sums of ds over the batch for 64 columns (the ones column of k_score_bwd's role A
//                               costs a fifth d tile at D = 256), Adagrad-scaled like role A's epilogue
#ifndef G4R_BWD2_WPE
#define G4R_BWD2_WPE 4
#endif
__global__ __launch_bounds__(256, G4R_BWD2_WPE) void k_score_bwd2(const DevModel* __restrict__ mp, StepState* st, int nblkA, int nblkB, int ndt, int nrt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int M = c.M, B = m.B, D = m.Dtop, N = m.N, ld = m.ldSc, tid = threadIdx.x;
    const GAS float* h = m.hd[m.n_layers - 1];
    const GAS float* Sc = m.Sc;
    const GAS float* Wy = m.Wy;
    const float lr = m.lr;
    const bool generic = m.generic != 0;
    constexpr int TILE_FLOATS = 4 * 64 * 16;
    int* sIt = reinterpret_cast<int*>(smem + TILE_FLOATS);      // role A: items of the tile's 64 score columns; role B: of the slab
    GAS long long* trc = (G4R_DBGTILE(m) && blockIdx.x < 2048) ? G4R_DBGTILE(m) + 8 * (size_t)(4096 + 2048 + blockIdx.x) : nullptr;
    if (trc && tid == 0) { trc[0] = wall_clock64(); trc[5] = c.t; trc[6] = (int)blockIdx.x < nblkA ? 0 : ((int)blockIdx.x < nblkA + nblkB ? 1 : 2); }
    if ((int)blockIdx.x < nblkA) {
        const int tile = G4R_XCD_TILE(blockIdx.x, nblkA);
        const int nt = tile / ndt, dt = tile - nt * ndt;
        const int n0 = nt * 64, d0 = dt * 64;
        if (tid < 64) sIt[tid] = (n0 + tid < N) ? m.col_item[n0 + tid] : -1;
        __syncthreads();
        auto aptr = [&](int kk, int kr, int cc) -> const GAS float* {       // ds[b = kk + kr][n0 + cc ..]
            return (kk + kr < M && n0 + cc < ld) ? Sc + (size_t)(kk + kr) * ld + n0 + cc : nullptr;
        };
        auto bptr = [&](int kk, int kr, int cc) -> const GAS float* {       // h[b = kk + kr][d0 + cc ..]
            return (kk + kr < M) ? h + (size_t)(kk + kr) * D + d0 + cc : nullptr;
        };
        GAS float* accWy = m.accWy;
        const GAS int* occ_fl = m.occ_fl;
        GAS float *dSy = G4R_DSY(m, c.g), *dAy = m.dAy;
        auto pre = [&](int n, int d) -> float4 {      // (accumulator in place for single-occurrence items: see k_score_bwd)
            const int item = sIt[n - n0];
            const bool ok = item >= 0;
            const int cnt = occ_fl[4 * (size_t)max(item, 0) + 2];
            return make_float4(ldf_at(accWy, (size_t)max(item, 0) * D + d, ok), ok ? 1.f : 0.f, __int_as_float(cnt), 0.f);
        };
        auto epi = [&](int n, int d, float g, float4 p) {
            if (n >= N) return;
            const float an = p.x + G4R_MUT_ACC(g * g);
            float step = (p.y != 0.f) ? G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD)) : 0.f;
            if (generic) step = (p.y != 0.f) ? g : 0.f;
            dSy[(size_t)n * D + d] = step;
            if (!generic && p.y != 0.f && __float_as_int(p.z) == 1) accWy[(size_t)sIt[n - n0] * D + d] = an;
            else dAy[(size_t)n * D + d] = an;
        };
        if (trc && tid == 0) trc[1] = wall_clock64();
        gemm_tile2k<true, false>(n0, d0, M, aptr, bptr, m.zrow, pre, epi, smem, trc);
        return;
    }
    if ((int)blockIdx.x < nblkA + nblkB) {
        const int w = G4R_XCD_TILE(blockIdx.x - nblkA, nblkB);
        const int per_kc = nrt * ndt;
        const int kc = w / per_kc, rem = w - kc * per_kc, rt = rem / ndt, dt = rem - rt * ndt;
        const int kch = m.kch, m0 = rt * 64, d0 = dt * 64, kbeg = kc * kch;
        if (m0 >= M) return;
        for (int i = tid; i < kch; i += 256) sIt[i] = (kbeg + i < ld) ? m.col_item[kbeg + i] : -1;
        __syncthreads();
        GAS float* dhpart = m.dhpart;
        auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? Sc + (size_t)(m0 + r) * ld + kbeg : nullptr; };
        auto bptr = [&](int kk, int kr, int cc) -> const GAS float* {       // Wy[item of column kbeg + kk + kr][d0 + cc ..]
            const int item = sIt[min(kk + kr, kch - 1)];
            return (item >= 0 && kk + kr < kch) ? Wy + (size_t)item * D + d0 + cc : nullptr;
        };
        auto epi = [&](int b, int d, float v, float4) {
            if (b < M) dhpart[((size_t)kc * B + b) * D + d] = v;
        };
        if (trc && tid == 0) trc[1] = wall_clock64();
        gemm_tile2k<false, true>(m0, d0, min(kch, ld - kbeg), arow, bptr, m.zrow, NoPre(), epi, smem, trc);
        return;
    }
    // ---- role C: 64 columns, thread (column tid & 63, row group tid >> 6)
    {
        const int n0 = ((int)blockIdx.x - nblkA - nblkB) * 64, cl = tid & 63, grp = tid >> 6, n = n0 + cl;
        const bool nok = n < ld;
        float s = 0.f;
        for (int b0 = grp; b0 < M; b0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ldf_if(Sc, (size_t)min(b0 + 4 * u, M - 1) * ld + (nok ? n : 0), nok && b0 + 4 * u < M);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        smem[grp * 64 + cl] = s;
        __syncthreads();
        if (tid < 64 && n < N) {
            const float g = (smem[cl] + smem[64 + cl]) + (smem[128 + cl] + smem[192 + cl]);
            const int item = m.col_item[n];
            const bool ok = item >= 0;
            const int cnt = m.occ_fl[4 * (size_t)max(item, 0) + 2];
            const float an = ldf_at(m.accBy, max(item, 0), ok) + G4R_MUT_ACC(g * g);
            float step = ok ? G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD)) : 0.f;
            if (generic) step = ok ? g : 0.f;
            G4R_DSBY(m, c.g)[n] = step;
            if (!generic && ok && cnt == 1) m.accBy[item] = an; else m.dABy[n] = an;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GRU backward (no BPTT: H is a constant input, gru4rec.py:460-463,576), element-wise head:
//   dh = sum of split-K slabs (top layer) or the upper layer's dy ; hidden-dropout mask ;
//   dz = dh (c - H) ; dc = dh z ; da = dc act'(c) ; dz' = dz z (1 - z)      -> dV[:, 0:D] = da, dV[:, 2D:3D] = dz'
__global__ __launch_bounds__(256) void k_gru_bwd_pre(const DevModel* __restrict__ mp, StepState* st, int l) {
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int M = c.M, B = m.B, D = m.D[l], D3 = 3 * D;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * D) return;
    const int row = e / D, d = e - row * D;
    const size_t o = (size_t)row * D + d;
    // everything this thread needs is requested in one round trip: the gate values first, then the split-K slabs of
    // dh (up to 24 at a time, clamped slab index + 0/1 weight instead of a data-dependent trip count)
    const float hv = m.H[l][c.g & 1][o], zz = m.z[l][o], cc = m.c[l][o];
    const int ks = m.ksplit;
    float dh = 0.f;
    if (l == m.n_layers - 1) {
        const GAS float* pp = m.dhpart + o;
        const size_t ps = (size_t)B * D;
        for (int k0 = 0; k0 < ks; k0 += 24) {
            float v[24];
#pragma unroll
            for (int q = 0; q < 24; ++q) v[q] = pp[(size_t)min(k0 + q, ks - 1) * ps];
#pragma unroll
            for (int q = 0; q < 24; ++q) dh += (k0 + q < ks) ? v[q] : 0.f;      // fixed summation order
        }
    } else if (m.bbn[l + 1] > 0) {
        // the upper layer's dy arrives as K-slice partial sums of its k_gru_bwd_bw (<= 16 slices, one round trip, slice order)
        const int nsl = m.bbn[l + 1];
        const GAS float* pp = m.dyp + o;
        const size_t ps = (size_t)B * D;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = pp[(size_t)min(q, nsl - 1) * ps];
#pragma unroll
        for (int q = 0; q < 16; ++q) dh += (q < nsl) ? v[q] : 0.f;
    } else {
        dh = m.dyl[l][o];
    }
    if (m.drop_h > 0.f) dh *= drop_mult(m.seed, (unsigned)c.g, G4R_STREAM_DROP_HIDDEN + l, row, d, 1.0f - m.drop_h);
    const float dz = dh * (cc - hv), dc = dh * zz;
    m.dV[l][(size_t)row * D3 + d] = dc * act_bwd_from_out(m.hidden_act, m.ha_p0, m.ha_p1, cc);
    m.dV[l][(size_t)row * D3 + 2 * D + d] = dz * zz * (1.f - zz);
}

// dr' = (da Wh^T) * H * r (1 - r)  -> dV[:, D:2D]      (B provider reads Wh rows: B[k][n] = Wh[n][k])
// (NTH / BK as for k_gru_p2)
template <int NTH, int BK>
__global__ __launch_bounds__(NTH) void k_gru_bwd_a(const DevModel* __restrict__ mp, StepState* st, int l) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int M = c.M, D = m.D[l], D3 = 3 * D;
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    if (m0 >= M) return;
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float* Hcur = m.H[l][c.g & 1];
    GAS float* dV = m.dV[l];
    const GAS float* rl = m.r[l];
    auto aload = [&](int kk, int r, int cc) -> float4 {
        const int row = m0 + r, k = kk + cc;
        return ld4_if(dV, (size_t)row * D3 + k, row < M && k < D);
    };
    auto bload = [&](int kk, int r, int cc) -> float4 {
        const int n = n0 + r, k = kk + cc;
        return ld4_if(Wh, (size_t)n * D + k, n < D && k < D);
    };
    auto pre = [&](int row, int n) -> float4 {
        const bool ok = row < M && n < D;
        const size_t o = (size_t)row * D + n;
        return make_float4(ldf_at(rl, o, ok), ldf_at(Hcur, o, ok), 0.f, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M || n >= D) return;
        dV[(size_t)row * D3 + D + n] = v * p.y * p.x * (1.f - p.x);
    };
    gemm_tile<GT_BM, GT_BN, BK, false, true, NTH>(m0, n0, D, aload, bload, pre, epi, smem);
}

// dy = dV Wx^T -> embedding-row gradient dSx (layer 0, through the embedding-dropout mask) or the lower layer's dh
__global__ __launch_bounds__(GT_NTH_FEW) void k_gru_bwd_b(const DevModel* __restrict__ mp, StepState* st, int l) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int M = c.M, D = m.D[l], IN = m.IN[l], D3 = 3 * D;
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    if (m0 >= M) return;
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* dV = m.dV[l];
    int* sRow = reinterpret_cast<int*>(smem + TileCfg<GT_BM, GT_BN, BB_BK, false, true>::SMEM_FLOATS);
    if (threadIdx.x < GT_BM) sRow[threadIdx.x] = (l == 0 && m0 + threadIdx.x < M) ? m.occ_idx[m0 + threadIdx.x] : -1;
    __syncthreads();
    auto aload = [&](int kk, int r, int cc) -> float4 {
        const int row = m0 + r, k = kk + cc;
        return ld4_if(dV, (size_t)row * D3 + k, row < M && k < D3);
    };
    auto bload = [&](int kk, int r, int cc) -> float4 {
        const int n = n0 + r, k = kk + cc;
        return ld4_if(Wx, (size_t)n * D3 + k, n < IN && k < D3);
    };
    GAS float* accT = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.accWy : m.accE;
    const GAS int* occ_fl = m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED) ? (size_t)0 : (size_t)m.n_items);
    const float lr = m.lr, drop_e = m.drop_e;
    const bool generic = m.generic != 0;
    const unsigned long long seed = m.seed;
    GAS float *dSx = G4R_DSX(m, c.g), *dAx = m.dAx, *dylo = (l > 0) ? m.dyl[l - 1] : nullptr;
    auto pre = [&](int row, int n) -> float4 {      // pre-step accumulator of the input item's row (layer 0) and its occurrence count
        const int item = (row - m0 < GT_BM) ? sRow[row - m0] : -1;
        const int cnt = occ_fl[4 * (size_t)max(item, 0) + 2];
        return make_float4(ldf_at(accT, (size_t)max(item, 0) * IN + n, item >= 0 && n < IN), __int_as_float(cnt), 0.f, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M || n >= IN) return;
        if (l == 0) {
            if (drop_e > 0.f) v *= drop_mult(seed, (unsigned)c.g, G4R_STREAM_DROP_EMBED, row, n, 1.0f - drop_e);
            const float an = p.x + G4R_MUT_ACC(v * v);
            dSx[(size_t)row * IN + n] = generic ? v : G4R_MUT_STEP(lr * v * frsq(an + G4R_EPS_ADAGRAD));
            // single-occurrence item: new accumulator in place (see k_score_bwd), else through dA and the update kernel's owner wave
            if (!generic && __float_as_int(p.y) == 1 && sRow[row - m0] >= 0) accT[(size_t)sRow[row - m0] * IN + n] = an;      // (item >= 0: the count read for a negative id is item 0's)
            else dAx[(size_t)row * IN + n] = an;
        } else {
            dylo[(size_t)row * IN + n] = v;
        }
    };
    GAS long long* clk = (G4R_DBGCLK(m) && blockIdx.x == 1 && blockIdx.y == 1) ? G4R_DBGCLK(m) + 16 : nullptr;     // kernel 1 of tools/clk.py
    gemm_tile<GT_BM, GT_BN, BB_BK, false, true, GT_NTH_FEW>(m0, n0, D3, aload, bload, pre, epi, smem, clk);
}

// ---------------------------------------------------------------------------------------------
// GRU backward of one layer in ONE launch, for layers of up to BF_MAXD units: replaces k_gru_bwd_pre + k_gru_bwd_a +
// k_gru_bwd_b (two dispatches less per layer and step).  One 8-wave workgroup per 16 x 32 tile of dy; everything it needs
// is requested up front (one round trip), the three stages then hand their results over through LDS:
//   stage 0  da = dh z act'(c), dz' = dh (c - H) z (1 - z) for the tile's 16 rows; dh = split-K slabs of k_score_bwd summed
//            in fixed order (top layer) or the upper layer's dy, through the hidden-dropout mask
//   stage 1  dr' = (da Wh^T) * H * r (1 - r), 16 rows x all D columns: one 16 x 16 sub-tile per wave (MFMA, Wh in LDS)
//   stage 2  dy tile = [da | dr' | dz'] Wx^T: two sub-tiles x four quarters of K = 3D over the eight waves (MFMA), partial
//            sums joined through LDS, epilogue of k_gru_bwd_b
// dV = [da | dr' | dz'] goes to memory from column tile 0 (the dense-gradient tiles read it).  The column tiles of a row
// block repeat stages 0 / 1 (a 16 x D x D product): cheaper than a launch boundary.  Thread -> element maps are powers of
// two (no integer divisions; this kernel is bound by instruction issue on the few CUs it occupies).
#define BF_MAXD 112
#define BF_ROWS 16
#define BF_SLB 10      // split-K slabs summed per batch of loads
__global__ __launch_bounds__(512) void k_gru_bwd_fused(const DevModel* __restrict__ mp, StepState* st, int l) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = c.M, B = m.B, D = m.D[l], IN = m.IN[l], D3 = 3 * D, Dq = D >> 2, D3q = D3 >> 2;
    const int m0 = blockIdx.y * BF_ROWS, n0 = blockIdx.x * 32;
    GAS long long* clk = (G4R_DBGCLK(m) && blockIdx.x == 1 && blockIdx.y == 1) ? G4R_DBGCLK(m) + 16 : nullptr;     // kernel 1 of tools/clk.py
    if (clk && tid == 0) clk[0] = wall_clock64();
    const int LDV = D3 + 2, LDW = D + 2;      // row strides with ld / 2 odd: MFMA fragment reads are conflict-free
    float* sV = smem;                          // [16][LDV]   dV rows of the tile
    float* sWh = sV + BF_ROWS * LDV;           // [D][LDW]    Wh[n][k]
    float* sWx = sWh + D * LDW;                // [32][LDV]   Wx[n0 + n][k]
    int* sRow = reinterpret_cast<int*>(sWx + 32 * LDV);
    f32x4* sR = reinterpret_cast<f32x4*>(smem + ((BF_ROWS * LDV + D * LDW + 32 * LDV + 32 + 3) & ~3));     // [6][64] partial sums
    const bool top = (l == m.n_layers - 1), writer = (blockIdx.x == 0);
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float *zl = m.z[l], *cl = m.c[l], *rl = m.r[l];
    GAS float* dV = m.dV[l];
    // ---- requests (clamped addresses, no branches in between).  The weight tiles do not depend on the step context: they are
    // requested before its first use, so that the state's memory round trip runs next to them instead of in front of them
    // Wh: 16 rows per pass, one quad of k per thread (32 quad slots per row, Dq <= 28 used)
    constexpr int NP_WH = (BF_MAXD + 15) / 16, NP_WX = (3 * BF_MAXD / 4 + 15) / 16;
    const int wr = tid >> 5, wq = min(tid & 31, Dq - 1);
    float4 wh[NP_WH], wx[NP_WX];
#pragma unroll
    for (int p = 0; p < NP_WH; ++p) wh[p] = ld4(Wh + (size_t)min(wr + 16 * p, D - 1) * D + 4 * wq);
    // Wx rows of the tile: 32 rows x 16 quad slots per pass
    const int xr = tid >> 4, xq = tid & 15;
    const GAS float* wxrow = Wx + (size_t)min(n0 + xr, IN - 1) * D3;
#pragma unroll
    for (int p = 0; p < NP_WX; ++p) wx[p] = ld4(wxrow + 4 * min(xq + 16 * p, D3q - 1));
    if (m0 >= M) return;      // (register loads: nothing is left behind)
    const GAS float* Hcur = m.H[l][c.g & 1];
    int myrow = m.occ_idx[min(m0 + (tid & 15), max(M - 1, 0))];
    if (!(l == 0 && m0 + (tid & 15) < M)) myrow = -1;
    // stage-0 operands: 16 rows x 32 quad slots
    const int r0 = tid >> 5, q0 = tid & 31;
    const bool act0 = q0 < Dq;
    const size_t off0 = (size_t)min(m0 + r0, max(M - 1, 0)) * D + 4 * min(q0, Dq - 1);
    const int ks = top ? m.ksplit : 1;
    const GAS float* dsrc = (top ? m.dhpart : m.dyl[l]) + off0;
    const size_t ps = (size_t)B * D;
    const float4 h4 = ld4(Hcur + off0), z4 = ld4(zl + off0), c4 = ld4(cl + off0);
    float4 g4[BF_SLB];
    {
        const GAS float* pp = dsrc;
#pragma unroll
        for (int q = 0; q < BF_SLB; ++q) {      // slots past the last slab re-read it (weight 0 below)
            g4[q] = ld4(pp);
            if (q + 1 < ks) pp += ps;
        }
    }
    // stage-1 epilogue operands: r and H at this wave's sub-tile of dr' (columns 16 wid ..)
    const int NT1 = (D + 15) >> 4;
    float r1[4], h1[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int row = m0 + 4 * lg + rg, n = wid * 16 + li;
        const bool ok = row < M && n < D;
        r1[rg] = ldf_at(rl, (size_t)row * D + n, ok);
        h1[rg] = ldf_at(Hcur, (size_t)row * D + n, ok);
    }
    if (clk && tid == 0) clk[1] = wall_clock64();
    // ---- Wh / Wx to LDS (row stride == 2 mod 4: 8-byte stores)
    if (tid < BF_ROWS) sRow[tid] = myrow;
#pragma unroll
    for (int p = 0; p < NP_WH; ++p) {
        const int n = wr + 16 * p;
        if (n < D && (tid & 31) < Dq) {
            float2* d = reinterpret_cast<float2*>(sWh + n * LDW + 4 * wq);
            d[0] = make_float2(wh[p].x, wh[p].y); d[1] = make_float2(wh[p].z, wh[p].w);
        }
    }
#pragma unroll
    for (int p = 0; p < NP_WX; ++p) {
        const int k4 = xq + 16 * p;
        if (k4 < D3q) {
            const bool ok = n0 + xr < IN;
            float2* d = reinterpret_cast<float2*>(sWx + xr * LDV + 4 * k4);
            d[0] = ok ? make_float2(wx[p].x, wx[p].y) : make_float2(0.f, 0.f);
            d[1] = ok ? make_float2(wx[p].z, wx[p].w) : make_float2(0.f, 0.f);
        }
    }
    if (clk && tid == 0) clk[2] = wall_clock64();
    // ---- stage 0
    const float drop_h = m.drop_h, hp0 = m.ha_p0, hp1 = m.ha_p1;
    const int hact = m.hidden_act;
    const unsigned long long seed = m.seed;
    {
        const int row = m0 + r0;
        float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < BF_SLB; ++q) {      // fixed summation order
            const float w = (q < ks) ? 1.f : 0.f;
            dh.x = fmaf(w, g4[q].x, dh.x); dh.y = fmaf(w, g4[q].y, dh.y); dh.z = fmaf(w, g4[q].z, dh.z); dh.w = fmaf(w, g4[q].w, dh.w);
        }
        for (int k0 = BF_SLB; k0 < ks; k0 += BF_SLB) {      // more slabs than one batch holds (rare)
            float4 v[BF_SLB];
#pragma unroll
            for (int q = 0; q < BF_SLB; ++q) v[q] = ld4(dsrc + (size_t)min(k0 + q, ks - 1) * ps);
#pragma unroll
            for (int q = 0; q < BF_SLB; ++q) {
                const float w = (k0 + q < ks) ? 1.f : 0.f;
                dh.x = fmaf(w, v[q].x, dh.x); dh.y = fmaf(w, v[q].y, dh.y); dh.z = fmaf(w, v[q].z, dh.z); dh.w = fmaf(w, v[q].w, dh.w);
            }
        }
        if (act0) {
            const bool ok = row < M;
            if (drop_h > 0.f) {
                const float4 mk = drop_mult4(seed, (unsigned)c.g, G4R_STREAM_DROP_HIDDEN + l, row, q0, 1.0f - drop_h);
                dh.x *= mk.x; dh.y *= mk.y; dh.z *= mk.z; dh.w *= mk.w;
            }
            const float hh[4] = {h4.x, h4.y, h4.z, h4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, dd[4] = {dh.x, dh.y, dh.z, dh.w};
            float da[4], dzp[4], ad[4];
            if (hact == G4R_ACT_TANH) {      // the default, kept out of the per-element switch
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[j] = 1.0f - cc[j] * cc[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[j] = act_bwd_from_out(hact, hp0, hp1, cc[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dz = dd[j] * (cc[j] - hh[j]), dc = dd[j] * zz[j];
                da[j] = ok ? dc * ad[j] : 0.f;
                dzp[j] = ok ? dz * zz[j] * (1.f - zz[j]) : 0.f;
            }
            float2* pa = reinterpret_cast<float2*>(sV + r0 * LDV + 4 * q0);
            float2* pz = reinterpret_cast<float2*>(sV + r0 * LDV + 2 * D + 4 * q0);
            pa[0] = make_float2(da[0], da[1]); pa[1] = make_float2(da[2], da[3]);
            pz[0] = make_float2(dzp[0], dzp[1]); pz[1] = make_float2(dzp[2], dzp[3]);
            if (writer && ok) {
                st4(dV + (size_t)row * D3 + 4 * q0, make_float4(da[0], da[1], da[2], da[3]));
                st4(dV + (size_t)row * D3 + 2 * D + 4 * q0, make_float4(dzp[0], dzp[1], dzp[2], dzp[3]));
            }
        }
    }
    if (clk && tid == 0) clk[3] = wall_clock64();
    __syncthreads();
    if (clk && tid == 0) clk[4] = wall_clock64();
    // stage-2 epilogue operands (waves 0, 1): pre-step accumulator of the input item's row, layer 0 (as k_gru_bwd_b)
    GAS float* accT = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.accWy : m.accE;
    const GAS int* occ_flT = m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED) ? (size_t)0 : (size_t)m.n_items);
    const int ns2 = wid & 1, kq = wid >> 1;
    float a2[4];
    int cnt2[4];      // occurrences of the row's item among the step's gathered rows (1: accumulator written in place below)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int item = sRow[4 * lg + rg], n = n0 + ns2 * 16 + li;
        a2[rg] = ldf_at(accT, (size_t)max(item, 0) * IN + n, item >= 0 && n < IN);
        cnt2[rg] = occ_flT[4 * (size_t)max(item, 0) + 2];
    }
    // k-steps in fully unrolled groups of 8 (all fragment reads ahead of the MFMAs); steps past kend read on inside the
    // workgroup's LDS and are replaced by zeros
    auto mma = [&](f32x4 acc, const float* pa, const float* pb, int kbeg, int kend) -> f32x4 {
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            float af[8], bf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 4 * u;
                const float a = pa[k], b = pb[k];
                af[u] = (k < kend) ? a : 0.f;
                bf[u] = (k < kend) ? b : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma16(af[u], bf[u], acc);
        }
        return acc;
    };
    // ---- stage 1: dr' for the tile's rows, sub-tile `wid`
    if (wid < NT1) {      // wave-uniform
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = mma(acc, sV + li * LDV + lg, sWh + (wid * 16 + li) * LDW + lg, 0, D);
        const int n = wid * 16 + li;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int r = 4 * lg + rg, row = m0 + r;
            const float v = acc[rg] * h1[rg] * r1[rg] * (1.f - r1[rg]);
            if (n < D) {
                sV[r * LDV + D + n] = (row < M) ? v : 0.f;
                if (writer && row < M) dV[(size_t)row * D3 + D + n] = v;
            }
        }
    }
    if (clk && tid == 0) clk[5] = wall_clock64();
    __syncthreads();
    if (clk && tid == 0) clk[6] = wall_clock64();
    // ---- stage 2: the dy tile; wave w: sub-tile (w & 1), quarter (w >> 1) of K = 3D
    const int kquart = ((D3q + 3) >> 2) << 2;
    f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2 = mma(acc2, sV + li * LDV + lg, sWx + (ns2 * 16 + li) * LDV + lg, kq * kquart, min(D3, (kq + 1) * kquart));
    if (clk && tid == 0) clk[7] = wall_clock64();
    if (kq) sR[(wid - 2) * 64 + lane] = acc2;
    __syncthreads();
    if (kq) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) {      // quarters 1..3 in order
        const f32x4 o = sR[(2 * j + ns2) * 64 + lane];
        acc2[0] += o[0]; acc2[1] += o[1]; acc2[2] += o[2]; acc2[3] += o[3];
    }
    const float lr = m.lr, drop_e = m.drop_e;
    const bool generic = m.generic != 0;
    GAS float *dSx = G4R_DSX(m, c.g), *dAx = m.dAx, *dylo = (l > 0) ? m.dyl[l - 1] : nullptr;
    const int n = n0 + ns2 * 16 + li;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int row = m0 + 4 * lg + rg;
        if (row >= M || n >= IN) continue;
        float v = acc2[rg];
        if (l == 0) {
            if (drop_e > 0.f) v *= drop_mult(seed, (unsigned)c.g, G4R_STREAM_DROP_EMBED, row, n, 1.0f - drop_e);
            const float an = a2[rg] + G4R_MUT_ACC(v * v);
            dSx[(size_t)row * IN + n] = generic ? v : G4R_MUT_STEP(lr * v * frsq(an + G4R_EPS_ADAGRAD));
            if (!generic && cnt2[rg] == 1 && sRow[4 * lg + rg] >= 0) accT[(size_t)sRow[4 * lg + rg] * IN + n] = an;      // single occurrence: in place (see k_score_bwd)
            else dAx[(size_t)row * IN + n] = an;
        } else {
            dylo[(size_t)row * IN + n] = v;
        }
    }
    if (clk && tid == 0) clk[8] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// Dense gradients: contractions over the batch, one wave per 16x16 output tile of
//   dWx = yin^T dV ; dWh = (H r)^T dV[:, :D] ; dWrz = H^T dV[:, D:] ; dBh = colsum(dV)
// with the dense Adagrad(+momentum) update (gru4rec.py:330-334,390-406) fused into the epilogue when
// no all-reduce is needed (single GPU); otherwise the gradient goes to dense_g for RCCL.
// One 16x16 output tile of a dense GRU gradient, resolved on the host: out[r0.., c0..] (leading dim ldo, at
// float offset `base` of the flat dense buffers) = X^T[., batch] * dV[batch, coff + .] ; X0/X1 = operand for
// even/odd global step (the hidden state ping-pongs) ; X == nullptr selects the bias row (column sums of dV).
// One-hot input (gru4rec.py:457-470): the layer-0 "input rows" are rows of Wx[0] itself, so their gradient is dV of
// layer 0 as it stands (no dy GEMM, no embedding dropout).  This turns it into the per-occurrence Adagrad step and
// new accumulator rows for k_sparse_update, like the epilogue of k_gru_bwd_b does for E / Wy rows.
__global__ __launch_bounds__(256) void k_onehot_step(const DevModel* __restrict__ mp, StepState* st) {
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int W = m.Ein, nc4 = W >> 2;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(q / nc4), c4 = (int)(q % nc4);
    if (row >= c.M) return;
    const int item = m.occ_idx[row];
    if (item < 0) return;      // (g4r_set_plan refuses ids outside the catalogue in active rows; belt and braces)
    const float lr = m.lr;
    const float4 g = ld4(m.dV[0] + (size_t)row * W + 4 * c4);
    const float4 a = ld4(m.accE + (size_t)item * W + 4 * c4);
    const int cnt = m.occ_fl[4 * ((size_t)m.n_items + item) + 2];
    const float4 an = make_float4(a.x + g.x * g.x, a.y + g.y * g.y, a.z + g.z * g.z, a.w + g.w * g.w);
    if (!m.generic && cnt == 1) st4(m.accE + (size_t)item * W + 4 * c4, an);      // single occurrence: in place (see k_score_bwd)
    else st4(m.dAx + (size_t)row * W + 4 * c4, an);
    if (m.generic) { st4(G4R_DSX(m, c.g) + (size_t)row * W + 4 * c4, g); return; }
    st4(G4R_DSX(m, c.g) + (size_t)row * W + 4 * c4, make_float4(lr * g.x * frsq(an.x + G4R_EPS_ADAGRAD), lr * g.y * frsq(an.y + G4R_EPS_ADAGRAD),
                                                         lr * g.z * frsq(an.z + G4R_EPS_ADAGRAD), lr * g.w * frsq(an.w + G4R_EPS_ADAGRAD)));
}

struct DenseTile {
    GP(const float) X0; GP(const float) X1; GP(const float) dV;
    long long base;
    int ldx, ldv, nrows, ncols, coff, ldo, r0, c0;
    int gather, pad;     // gather = 1: X rows are the step's input embedding rows table[in_idx[b]] (+ embedding dropout)
};

__device__ __forceinline__ void dense_adagrad(const DevModel& m, size_t off, float g) {
    const float acc = m.dense_acc[off] + g * g;
    m.dense_acc[off] = acc;
    const float gs = g * frsq(acc + G4R_EPS_ADAGRAD);
    const float p = m.dense_p[off];
    if (m.mom > 0.f) {
        const float v = m.mom * m.dense_vel[off] - m.lr * (gs + m.lmbd * p);
        m.dense_vel[off] = v;
        m.dense_p[off] = p + v;
    } else {
        m.dense_p[off] = p * (1.0f - m.lr * m.lmbd) - m.lr * gs;
    }
}

// One workgroup per 32x32 output tile of a dense GRU gradient (contraction over the batch):
//   dWx = yin^T dV ; dWh = (H r)^T dV[:, :D] ; dWrz = H^T dV[:, D:] ; dBh = colsum(dV)
// with the dense Adagrad(+momentum) update (gru4rec.py:330-334,390-406) fused into the epilogue when no all-reduce
// is needed (single GPU); otherwise the gradient goes to dense_g for RCCL.  Layer-0 input rows come from yin0
// (published by k_gru_p1), never from the embedding table, so this may run next to the sparse update.
template <int DT>
__device__ __forceinline__ void dense_grad_tile(const DevModel& m, StepState* st, const DenseTile* tiles_, int tile, float* smem) {
    const GAS DenseTile* tiles = (const GAS DenseTile*)tiles_;   // same mangled signature on both passes
    const StepCtx c = load_ctx(st);
    const DenseTile tl = tiles[tile];    // fully resolved on the host: no per-layer lookups here
    const GAS float* X = tl.gather ? (const GAS float*)m.yin0 : ((c.g & 1) ? tl.X1 : tl.X0);
    const GAS float* dV = tl.dV;
    const int M = c.M;
    const bool ones = (X == nullptr);          // bias row: column sums of dV
    const float lr = m.lr, momc = m.mom, lmbd = m.lmbd;
    const int inplace = m.apply_dense_inplace;
    GAS float *dp = m.dense_p, *dacc = m.dense_acc, *dvel = m.dense_vel, *dg = m.dense_g;
    auto aload = [&](int kk, int r, int cc) -> float4 {      // staging tile [k = b][m = output row]
        const int b = kk + r, rr = tl.r0 + cc;
        const bool ok = b < M && rr < tl.nrows;
        if (ones) return make_float4((ok && rr == 0) ? 1.f : 0.f, 0.f, 0.f, 0.f);
        return ld4_if(X, (size_t)b * tl.ldx + rr, ok);
    };
    auto bload = [&](int kk, int r, int cc) -> float4 {
        const int b = kk + r, col = tl.c0 + cc;
        return ld4_if(dV, (size_t)b * tl.ldv + tl.coff + col, b < M && col < tl.ncols);
    };
    auto pre = [&](int row, int col) -> float4 {      // optimizer state of the element (accumulator, parameter, velocity)
        const bool ok = inplace && row < tl.nrows && col < tl.ncols;
        const size_t off = (size_t)tl.base + (size_t)row * tl.ldo + col;
        return make_float4(ldf_at(dacc, off, ok), ldf_at(dp, off, ok), ldf_at(dvel, off, ok && momc > 0.f), 0.f);
    };
    auto epi = [&](int row, int col, float g, float4 p) {
        if (row >= tl.nrows || col >= tl.ncols) return;
        const size_t off = (size_t)tl.base + (size_t)row * tl.ldo + col;
        if (!inplace) { dg[off] = g; return; }
        const float acc = p.x + G4R_MUT_DACC(g * g);            // gru4rec.py:330-334,390-406
        dacc[off] = acc;
        const float gs = g * frsq(acc + G4R_EPS_ADAGRAD);
        if (momc > 0.f) {
            const float v = momc * p.z - lr * (gs + lmbd * p.y);
            dvel[off] = v;
            dp[off] = p.y + v;
        } else {
            dp[off] = p.y * (1.0f - lr * lmbd) - lr * gs;
        }
    };
    static_assert(DT == 32, "tile edge");
    gemm_tile<GT_BM, GT_BN, GT_BK, true, false, GT_NTH_FEW>(tl.r0, tl.c0, M, aload, bload, pre, epi, smem);
}
template <int DT>
__global__ __launch_bounds__(GT_NTH_FEW) void k_dense_grad(const DevModel* __restrict__ mp, StepState* st, const DenseTile* __restrict__ tiles_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    dense_grad_tile<DT>(*mp, st, tiles_, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Generic optimizer path (adapt != adagrad or grad_cap > 0; gru4rec.py:300-381,386-432).  For one parameter element that
// received n gradients g_1..g_n this step (dense: n = 1): S = sum g_i, Q = sum g_i^2, T1 = sum g_i / sqrt(a0 + g_i^2 + eps)
// (adagrad only), gk = the last one.  Returns the summed scaled gradient G, the scaled last gradient gl (momentum) and the
// new statistics.  `dense` selects Adam's proper first moment; its sparse branch feeds grad**2 into the mean (:325), and
// both bias corrections use beta1 (:329) -- reproduced.
struct OptOut { float G, gl, A, U, C; };
__device__ __forceinline__ OptOut opt_rule(int adapt, float v1, float v3, bool dense, float a0, float u0, float c0, float S, float Q,
                                           float T1, float gk, float fn) {
    OptOut o;
    o.U = u0; o.C = c0;
    const float eps = G4R_EPS_ADAGRAD;
    if (adapt == G4R_ADAPT_RMSPROP) {
        const float an = v1 * a0 + (1.f - v1) * Q, sc = 1.f / sqrtf(an + eps);
        o.G = S * sc; o.gl = gk * sc; o.A = an;
    } else if (adapt == G4R_ADAPT_ADADELTA) {
        const float an = v1 * a0 + (1.f - v1) * Q, r = (u0 + eps) / (an + eps), sc = sqrtf(r);
        o.U = v1 * u0 + (1.f - v1) * r * Q;
        o.G = S * sc; o.gl = gk * sc; o.A = an;
    } else if (adapt == G4R_ADAPT_ADAM) {
        const float an = v3 * a0 + (1.f - v3) * Q, mn = v1 * u0 + (1.f - v1) * (dense ? S : Q), cn = c0 + 1.f;
        const float corr = 1.f - powf(v1, cn), out = (mn / corr) / (sqrtf(an / corr) + eps);
        o.G = fn * out; o.gl = out; o.A = an; o.U = mn; o.C = cn;
    } else if (adapt == G4R_ADAPT_NONE) {
        o.G = S; o.gl = gk; o.A = a0;
    } else {
        o.A = a0 + gk * gk;
        o.G = T1; o.gl = gk / sqrtf(o.A + eps);
    }
    return o;
}

// sum of squares of every gradient of the step (dense buffer + per-occurrence sparse rows), gru4rec.py:387
__global__ __launch_bounds__(256) void k_grad_sqsum(const DevModel* __restrict__ mp, StepState* st) {
    __shared__ float red[8];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const long long nx = (long long)c.M * m.Ein, ny = (long long)m.N * m.Dtop, nb = m.N, nd = m.dense_count;
    const long long total = nx + ny + nb + nd;
    float s = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)G4R_NORM_BLOCKS * 256) {
        float g;
        if (e < nx) g = m.dSx[e];
        else if (e < nx + ny) g = m.dSy[e - nx];
        else if (e < nx + ny + nb) g = m.dSBy[e - nx - ny];
        else g = m.dense_g[e - nx - ny - nb] * m.grad_scale;
        s += g * g;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) m.gsq_part[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_grad_clip(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    float s = 0.f;
    for (int i = threadIdx.x; i < G4R_NORM_BLOCKS; i += 64) s += m.gsq_part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        m.gclip[0] = (norm >= m.grad_cap) ? m.grad_cap / norm : 1.f;      // T.switch(T.ge(norm, cap), g * cap / norm, g)
    }
}

// after the RCCL all-reduce: element-wise dense rule on the averaged gradient (element i of the flat dense buffers)
__device__ __forceinline__ void dense_apply_elem(const DevModel& m, int i) {
    if (!m.generic) { dense_adagrad(m, (size_t)i, m.dense_g[i] * m.grad_scale); return; }
    float gs = 0.f;
    if (m.xmode != 0) {      // exact-replica mode: the ranks' raw gradients out of the all-gathered blocks, in rank order
        for (int q = 0; q < m.xn; ++q) gs += (m.xbase + (long long)q * m.xstride)[m.xoffDg + i];
    } else gs = m.dense_g[i];
    const float g = gs * m.grad_scale * m.gclip[0];
    const float a0 = m.dense_acc[i], u0 = m.dense_acc2 ? m.dense_acc2[i] : 0.f, c0 = m.dense_cnt ? m.dense_cnt[i] : 0.f;
    const OptOut o = opt_rule(m.adapt, m.ap0, m.ap1, true, a0, u0, c0, g, g * g, g / sqrtf(a0 + g * g + G4R_EPS_ADAGRAD), g, 1.f);
    m.dense_acc[i] = o.A;
    if (m.dense_acc2) m.dense_acc2[i] = o.U;
    if (m.dense_cnt) m.dense_cnt[i] = o.C;
    const float p = m.dense_p[i];
    if (m.mom > 0.f) {      // gru4rec.py:400-404
        const float v = m.mom * m.dense_vel[i] - m.lr * (o.G + m.lmbd * p);
        m.dense_vel[i] = v;
        m.dense_p[i] = p + v;
    } else {
        m.dense_p[i] = p * (1.0f - m.lr * m.lmbd) - m.lr * o.G;
    }
}
__global__ __launch_bounds__(256) void k_dense_apply(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m.dense_count) dense_apply_elem(m, i);
}

// ---------------------------------------------------------------------------------------------
// Sparse Adagrad(+momentum) on the gathered rows, gru4rec.py:335-340,407-431, with the reference's
// duplicate-index semantics made deterministic:
//   - every occurrence is scaled with the PRE-step accumulator: g~ = g / sqrt(acc_old + g^2 + eps)
//     (done by the gradient producers: dS* hold the scaled steps, dA* hold acc_old + g^2)
//   - parameter increments of duplicates accumulate (inc_subtensor)
//   - accumulator / velocity take the value of the LAST occurrence (set_subtensor, NumPy order)
// so for an item with n occurrences, S = sum of its step rows and s_k = the step row of its last occurrence k:
//   no momentum:  P = P0 - (S + n*reg)                      reg = lr*lmbd*P0
//   momentum:     P = P0 + n*mom*V0 - (S + n*reg) ,  V = mom*V0 - (s_k + reg)
//   A = dA[k]
// Items with ONE occurrence (count field of occ_fl == 1; ~80 % of a step's rows) have had their accumulator written in place by
// the gradient producer (k_score_bwd* / k_gru_bwd_* epilogues): for them this kernel moves THREE rows -- step row read, parameter
// row read + write -- and is done before the workgroup's first barrier.  Only items with several occurrences go through dA (read
// by the owner once the count is known) and the occurrence list.
// One wave per occurrence k of (X | Y | samples); the wave of an item's LAST occurrence owns the row, so the
// row state and s_k can be requested before anything is known about duplicates.  The occurrence list is
// staged in LDS once per workgroup and scanned with ballots.  Up to SP_UB earlier occurrences are summed by the
// owner in one batch of loads (one round trip); hotter items (popularity-sampled negatives repeat the head of
// the catalogue dozens of times per step) are summed by all SP_WAVES waves of the workgroup together, wave w taking
// every SP_WAVES-th occurrence, partial sums combined through LDS in wave order.  No atomics, bit-reproducible.
// The extra last block folds the per-row losses into loss_steps[t] and advances the step state.
#define SP_WAVES 8   // occurrences (waves) per workgroup
#ifndef SP_UB
#define SP_UB 4
#endif
#ifndef SP_HOT
#define SP_HOT 8
#endif
// SP_UB: float4 step-row chunks one lane fetches together (one round trip); SP_HOT: items with more earlier occurrences (in
// float4 chunks per lane) are "hot".  (Round 3 fetched SP_HOT chunks per round trip: 32 more registers at the peak of a path a
// fifth of the waves take, which cost the kernel its spills; 5-8 earlier occurrences now take two round trips.)

// MAXCH = float4 chunks per lane (1: row width <= 256, 2: <= 512, 4: <= 1024).  MOM: the model trains with momentum (velocity
// rows read and written; a compile-time switch: as a run-time one its condition mask was the last scalar register hipcc spilled).
template <int MAXCH, bool MOM>
__device__ __forceinline__ void sparse_update_block(const DevModel* __restrict__ mp, StepState* st, int nblk_occ, int blk, float* smem) {
    const DevModel& m = *mp;
    constexpr int UB = SP_UB / MAXCH, HOT = SP_HOT / MAXCH;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int B = m.B, R = m.R;
    // descriptor fields used inside loops are snapshotted into registers: re-reading them through `mp` costs a
    // scalar-memory round trip per iteration (the compiler does not hoist them across the global stores)
    const float lr = m.lr, momc = m.mom, lmbd = m.lmbd;
    const bool constrained = (m.embed_mode == G4R_EMBED_CONSTRAINED);
    // (the output-bias tables are NOT snapshotted: they are used twice, far apart, and holding six more scalar registers across
    // the whole kernel made hipcc spill eight of them to vector lanes)
    GAS float *tE = m.E, *tWy = m.Wy, *tvE = m.velE, *tvWy = m.velWy, *taE = m.accE, *taWy = m.accWy;
    const int wE = m.Ein, wY = m.Dtop;
    const GAS int* g_occ = m.occ_idx;
    // step planes of this step (a ring of slots when row updates may be deferred: DevModel::defer_mask; the step's global index is
    // read next to the wave's first loads -- *_b is not written by this launch)
    constexpr bool CAN_DEFER = !MOM;
    const int dmask = CAN_DEFER ? m.defer_mask : 0;
    const long long gq = dmask ? ((const GAS StepState*)st)->g_b : 0;
    const GAS float *g_dSx = G4R_DSX(m, gq), *g_dSy = G4R_DSY(m, gq), *g_dSBy = G4R_DSBY(m, gq);      // (the dA planes are only read by owners of repeated items: not snapshotted)
    if (blk == nblk_occ) {
        // ---- bookkeeping block: cost = sum_i L_i / batch_size (gru4rec.py:577), NaN flag (:626), advance state
        // (the only block of this role that needs the step context: the row update works from occ_idx / occ_fl alone)
        const StepCtx c = load_ctx(st);
        const int Mn = m.Mplan[c.t + 1];     // the plan carries one trailing entry (and one trailing row)
        if (wid == 0) {
            float s = 0.f;
            for (int i = lane; i < c.M; i += 64) s += m.lossrow[i];
            s = wave_sum(s);
            if (lane == 0) {
                const float cost = s * m.inv_B;
                m.loss_steps[c.t] = cost;
                GAS StepState* sg = (GAS StepState*)st;
                if (isnan(cost)) sg->nan_flag = 1;
                sg->t_a = c.t + 1;
                sg->g_a = c.g + 1;
                sg->M_a = Mn;
            }
        }
        stage_step_inputs(m, c.t + 1, c.g + 1, Mn, tid, SP_WAVES * 64);
        return;
    }
    const long long t_start = G4R_DBGCLK(m) ? wall_clock64() : 0;
    // LDS: occurrence list padded with -2 to a multiple of 256 (+256) | hot-item slots | per-wave match lists |
    // per-wave partial sums
    const int Rpad = ((R + 255) & ~255) + 256;
    const int PW = max(wE, wY) + 4;               // partial row: W floats + (bias partial, count, count among Y|samples, pad)
    int* sOcc = reinterpret_cast<int*>(smem);
    int* sHot = sOcc + Rpad;                      // [SP_WAVES] item, [SP_WAVES] first occurrence
    int* sList = sHot + 2 * SP_WAVES;             // [SP_WAVES][64]
    float* sPart = reinterpret_cast<float*>(sList + 64 * SP_WAVES);   // [SP_WAVES][PW]
    int* myList = sList + 64 * wid;
    GAS int* g_fl = m.occ_fl;
    const int nI = m.n_items;
    // occurrence of this wave: strided over the workgroups (wave w of workgroup b takes k = w * nblk + b).  The last occurrences of
    // the popular items -- their owners, which have the duplicate sums to do -- sit together at the end of the list; with a
    // contiguous mapping they would share a few workgroups that then run their hot-item rounds one after the other
    const int k = wid * nblk_occ + blk;
    // short occurrence lists (Rpad <= 4096) are requested right away, next to the first loads of the wave, and only written to
    // LDS if some wave turns out to own an item with earlier occurrences; longer lists are fetched when that is known (for
    // those the loads below all go to element 0: one cache line per wave, no branch between the loads)
    constexpr int EARLY = (MAXCH >= 2 && MOM) ? 1 : 2;      // (momentum at two chunks per lane: the second early quad is what would spill)
    const int n4 = Rpad >> 2;
    const GAS int4* g_occ4 = (const GAS int4*)g_occ;
    const bool early = n4 <= EARLY * SP_WAVES * 64;
    int4 ev[EARLY];
#pragma unroll
    for (int q = 0; q < EARLY; ++q) ev[q] = g_occ4[early ? min(q * SP_WAVES * 64 + tid, n4 - 1) : 0];
    int item = g_occ[min(k, R - 1)];
    // deferral candidate: this occurrence is its item's last use inside the current window of steps (k_defer_scan, from the plan and
    // the sample store: known ahead).  If it also is the item's ONLY occurrence of this step, nothing will gather the row before the
    // window's flush launch: the wave then moves no row at all -- the step row stays in its ring slot, the item goes to dlist.
    const size_t dslot = (size_t)(gq & dmask) * (size_t)m.dRcap + (size_t)min(k, m.dRcap - 1);
    const bool cand = dmask != 0 && k < R && m.dcand[dslot] != 0;      // wave-uniform
    if (k >= R) item = -1;
    // occurrence range sharing a table with k: constrained -> all of X|Y|samples ; separate -> X alone, Y|samples alone
    const int lo = (constrained || k < B) ? 0 : B;
    const bool tableE = (k < B && !constrained);
    GAS float* P = tableE ? tE : tWy;
    GAS float* A = tableE ? taE : taWy;
    GAS float* V = tableE ? tvE : tvWy;
    const int W = tableE ? wE : wY;
    const int nc4 = W >> 2;
    constexpr bool mom = MOM;
    const bool bias = (k >= B);
    // ---- the item's (last, first, count) entry (published with atomics by k_gru_p1 / k_score_fwd), the row state and
    // the last occurrence's step / accumulator rows: one round trip (unconditional loads with clamped indices:
    // what a non-owner fetches is simply not used)
    const int item_c = max(item, 0), k_c = min(k, R - 1);
    GAS int* flp = g_fl + 4 * ((tableE ? (size_t)nI : 0) + item_c);
    const int4 fl = ldi4(flp);
    const GAS float* srow_k = (k_c < B) ? g_dSx + (size_t)k_c * W : g_dSy + (size_t)(k_c - B) * W;
    float4 pz[MAXCH], vz[MAXCH], sk[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) { pz[q] = make_float4(0.f, 0.f, 0.f, 0.f); sk[q] = pz[q]; vz[q] = pz[q]; }
    auto load_rows = [&]() {
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int cc = 4 * min(lane + 64 * q, nc4 - 1);
            pz[q] = ld4(P + (size_t)item_c * W + cc);
            sk[q] = ld4(srow_k + cc);
            vz[q] = mom ? ld4(V + (size_t)item_c * W + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (!cand) load_rows();      // (a candidate waits for its entry: most candidates are deferred and never touch their rows)
    float bpz = 0.f, bvz = 0.f, bsk = 0.f;
    if (bias) {
        bpz = m.By[item_c]; bsk = g_dSBy[k_c - B];
        if (mom) bvz = m.velBy[item_c];
    }
    // the wave of the item's last occurrence owns the row (and clears the item's entry for the next step)
    const bool owner = item >= 0 && fl.x == k + 1;
    const int first_j = max(lo, R - fl.y);
    // An item ALL of whose occurrences are sampled negatives of this step (first occurrence >= 2B: the common kind of repeat, the
    // popularity sampler draws the head of the catalogue several times per row): its score columns are copies of one another --
    // same item row, same bias, no column of them is anybody's positive -- so k_loss_rows / k_score_bwd produced bit-identical
    // step rows for them and the sum over the earlier occurrences is (count - 1) x this wave's own row, added one at a time in
    // the order the list walk would have used: no occurrence list, no second round trip for the step rows.
    const bool deferred = cand && owner && fl.z == 1;      // wave-uniform
    if (dmask != 0 && lane == 0 && k < m.dRcap) m.dlist[dslot] = deferred ? item : -1;
    if (cand && !deferred) load_rows();      // a candidate that repeats inside its own step (or is not an owner): the usual path, one round trip later
    const bool allsmp = owner && fl.z > 1 && first_j >= 2 * B;
    const bool dup = owner && fl.z > 1 && !allsmp;
    const bool hot = owner && fl.z - 1 > HOT && !allsmp;
    if (owner && lane == 0) {
        *(GAS int4*)flp = make_int4(0, 0, 0, 0);
        if (m.touched) m.touched[(tableE ? (size_t)nI : 0) + item] = 1;
    }
    // final row values from the sum `ss` of the item's step rows, the last occurrence's step row `sl` and the pre-step row state;
    // n occurrences in all, nb of them among Y | samples (the output bias is only touched by those, gru4rec.py:486-489)
    auto finish = [&](const float4 (&S)[MAXCH], float Sb, int n, int nb) {
        const float fn = (float)n;
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int c4 = lane + 64 * q;
            const float p0[4] = {pz[q].x, pz[q].y, pz[q].z, pz[q].w}, v0[4] = {vz[q].x, vz[q].y, vz[q].z, vz[q].w};
            const float sl[4] = {sk[q].x, sk[q].y, sk[q].z, sk[q].w};
            const float ss[4] = {S[q].x + sk[q].x, S[q].y + sk[q].y, S[q].z + sk[q].z, S[q].w + sk[q].w};
            float pn[4], vn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float reg = (lmbd > 0.f) ? lr * lmbd * p0[e] : 0.f;
                const float tot = (lmbd > 0.f) ? ss[e] + fn * reg : ss[e];
                if (mom) { vn[e] = momc * v0[e] - (sl[e] + reg); pn[e] = p0[e] + (fn * (momc * v0[e]) - tot); }
                else { vn[e] = 0.f; pn[e] = p0[e] - tot; }
            }
            if (c4 < nc4) {
                const size_t o = (size_t)item * W + 4 * c4;
                st4(P + o, make_float4(pn[0], pn[1], pn[2], pn[3]));
                if (mom) st4(V + o, make_float4(vn[0], vn[1], vn[2], vn[3]));
            }
        }
        if (bias && lane == 0) {
            const float fb = (float)nb;
            const float reg = (lmbd > 0.f) ? lr * lmbd * bpz : 0.f;
            const float sb = Sb + bsk;
            const float tot = (lmbd > 0.f) ? sb + fb * reg : sb;
            if (mom) { m.By[item] = bpz + (fb * (momc * bvz) - tot); m.velBy[item] = momc * bvz - (bsk + reg); }
            else m.By[item] = bpz - tot;
        }
    };
    float4 S[MAXCH];
    float Sb = 0.f;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) S[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- the common case: the item's only occurrence.  Its accumulator is already in place (written by the producer of the step
    // row); parameter (and velocity) rows are final right here, ahead of the workgroup's barrier
    const bool single = owner && fl.z == 1;
    if (single && !deferred) finish(S, 0.f, 1, bias ? 1 : 0);
    // owners of items with several occurrences: the last occurrence's accumulator row (dA plane), requested now that the count
    // is known -- it lands during the barrier / the list walk below
    float4 ak[MAXCH];
    float bak = 0.f;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) ak[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (owner && !single) {      // wave-uniform
        const GAS float* arow_k = (k_c < B) ? m.dAx + (size_t)k_c * W : m.dAy + (size_t)(k_c - B) * W;
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) ak[q] = ld4(arow_k + 4 * min(lane + 64 * q, nc4 - 1));
        if (bias) bak = m.dABy[k_c - B];
    }
    const long long t_own = G4R_DBGCLK(m) ? wall_clock64() : 0;

    // scan of sOcc[a, b) for `it`: match number i (ascending) goes to myList[i - 64 * pass]; returns the
    // number of matches, nb = those among Y|samples
    auto scan = [&](int it, int a, int b, int pass, int& nb) {
        int idx = 0;
        nb = 0;
        constexpr int NV = (MAXCH >= 2 && MOM) ? 2 : 4;      // (momentum at two chunks per lane: the register budget is at its limit)
        for (int base0 = a & ~255; base0 < b; base0 += 256 * NV) {
            // 1024 entries per step: four 16-byte LDS reads per lane (four consecutive entries each) are requested together;
            // reads past Rpad stay inside the workgroup's LDS allocation and can never match (j < b fails)
            int4 vv[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) vv[u] = *reinterpret_cast<const int4*>(sOcc + base0 + 256 * u + 4 * lane);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
            const int4 v = vv[u];
            const int j = base0 + 256 * u + 4 * lane;
            const bool h0 = v.x == it && j >= a && j < b, h1 = v.y == it && j + 1 >= a && j + 1 < b;
            const bool h2 = v.z == it && j + 2 >= a && j + 2 < b, h3 = v.w == it && j + 3 >= a && j + 3 < b;
            if (__ballot(h0 || h1 || h2 || h3) == 0) continue;       // the common case: a few compares and a scalar branch
            // ascending occurrence order = lane-major: all matches of lower lanes first, then this lane's earlier elements
            const bool hh[4] = {h0, h1, h2, h3};
            unsigned long long mk[4];
            int below = 0, total = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mk[e] = __ballot(hh[e]);
                below += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[e], 0u));
                total += __popcll(mk[e]);
                nb += __popcll(__ballot(hh[e] && j + e >= B));
            }
            int ord = idx - 64 * pass + below;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (hh[e]) {
                    if (ord >= 0 && ord < 64) myList[ord] = j + e;
                    ++ord;
                }
            }
            idx += total;
            }
        }
        return idx;
    };

    // S = sum of the step rows of the item's occurrences before k, Sb = the same for the output bias
    int n_e = 0, nb_e = 0;
    if (lane == 0) { sHot[wid] = hot ? item : -1; sHot[SP_WAVES + wid] = first_j; }
    const bool any_dup = __syncthreads_or(dup ? 1 : 0) != 0;
    long long t_col = t_own, t_app = t_own, t_h[5] = {0, 0, 0, 0, 0};      // t_h: phases of the last hot round (debug)
    if (any_dup) {
        // ---- some wave of this workgroup owns an item with earlier occurrences: stage the occurrence list
        auto commit4 = [&](int j4, int4 v) {
            const int j = 4 * j4;
            if (j4 < n4) *reinterpret_cast<int4*>(sOcc + j) = make_int4(j < R ? v.x : -2, j + 1 < R ? v.y : -2, j + 2 < R ? v.z : -2, j + 3 < R ? v.w : -2);
        };
        if (early) {
#pragma unroll
            for (int q = 0; q < EARLY; ++q) commit4(q * SP_WAVES * 64 + tid, ev[q]);
        } else {
            // 16-byte loads, up to 4 in flight per thread: one round trip for R <= 8192 (the buffer is padded to Rpad ints)
            for (int j0 = 0; j0 < n4; j0 += 4 * SP_WAVES * 64) {
                int4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = g_occ4[min(j0 + q * SP_WAVES * 64 + tid, n4 - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) commit4(j0 + q * SP_WAVES * 64 + tid, v[q]);
            }
        }
        __syncthreads();
        if (G4R_DBGCLK(m)) t_col = wall_clock64();
        // one code path for both kinds of work (rarely executed code is instruction-cache cold, so it is kept small):
        //   h = -1 : a wave sums the (<= UB) earlier occurrences of its own item, range [first, k)
        //   h >= 0 : hot item of wave h; every wave sums the occurrences found in its slice of [first, k_h), the
        //            partial sums are combined through LDS in wave (= occurrence) order
        // hot owners of this workgroup (one LDS read instead of one per candidate wave)
        unsigned hm = (unsigned)__ballot(lane < SP_WAVES && sHot[lane & (SP_WAVES - 1)] >= 0);
        for (int h = -1; h < SP_WAVES; h = hm ? (int)__builtin_ctz(hm) : SP_WAVES, hm &= hm - 1) {
            int it = item, a = first_j, b = k, tW = W, tnc4 = nc4;
            bool tb = bias, active = dup && !hot;
            if (h >= 0) {
                it = sHot[h];
                if (it < 0) continue;             // workgroup-uniform
                const int hk = h * nblk_occ + blk, hlo = sHot[SP_WAVES + h];
                const int slice = (((hk - hlo + SP_WAVES - 1) / SP_WAVES) + 63) & ~63;
                a = hlo + wid * slice; b = min(hk, a + slice);
                tW = (hk < B && !constrained) ? wE : wY; tnc4 = tW >> 2;
                tb = hk >= B; active = true;
            }
            float4 T[MAXCH];
            float Tb = 0.f;
            int n_w = 0, nb_w = 0;
            if (G4R_DBGCLK(m) && h >= 0) t_h[0] = wall_clock64();
#pragma unroll
            for (int q = 0; q < MAXCH; ++q) T[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active) {
                for (int pass = 0;; ++pass) {
                    if (h < 0 && fl.z == 2) {          // one earlier occurrence: it is the first one, nothing to search
                        if (lane == 0) myList[0] = a;
                        n_w = 1; nb_w = (a >= B) ? 1 : 0;
                    } else {
                        n_w = scan(it, a, b, pass, nb_w);
                    }
                    if (G4R_DBGCLK(m) && h >= 0 && pass == 0) t_h[1] = wall_clock64();
                    const int cnt = min(n_w - 64 * pass, 64);
                    const int myj = lane < cnt ? myList[lane] : -1;
                    const float bd = (tb && myj >= B) ? g_dSBy[max(myj - B, 0)] : 0.f;
                    for (int i0 = 0; i0 < cnt; i0 += UB) {        // UB step rows per round trip
                        float4 g[UB][MAXCH];
                        float w[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            // branch-free: slots past the end re-read the last row with weight 0
                            w[u] = (i0 + u < cnt) ? 1.f : 0.f;
                            const int jj = __builtin_amdgcn_readlane(myj, min(i0 + u, cnt - 1) & 63);
                            const GAS float* srow = (jj < B) ? g_dSx + (size_t)jj * tW : g_dSy + (size_t)(jj - B) * tW;
#pragma unroll
                            for (int q = 0; q < MAXCH; ++q) g[u][q] = ld4(srow + 4 * min(lane + 64 * q, tnc4 - 1));
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
#pragma unroll
                            for (int q = 0; q < MAXCH; ++q) {
                                T[q].x = fmaf(w[u], g[u][q].x, T[q].x); T[q].y = fmaf(w[u], g[u][q].y, T[q].y);
                                T[q].z = fmaf(w[u], g[u][q].z, T[q].z); T[q].w = fmaf(w[u], g[u][q].w, T[q].w);
                            }
                        }
                    }
                    if (tb) Tb += wave_sum(bd);
                    if (n_w <= 64 * (pass + 1)) break;
                }
            }
            if (h < 0) {
                if (active) {
#pragma unroll
                    for (int q = 0; q < MAXCH; ++q) S[q] = T[q];
                    Sb = Tb; n_e = n_w; nb_e = nb_w;
                }
                if (G4R_DBGCLK(m)) t_app = wall_clock64();
                continue;
            }
            if (G4R_DBGCLK(m)) t_h[2] = wall_clock64();
            float* part = sPart + wid * PW;
#pragma unroll
            for (int q = 0; q < MAXCH; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < tnc4) *reinterpret_cast<float4*>(part + 4 * c4) = T[q];
            }
            if (lane == 0) { part[PW - 4] = Tb; part[PW - 3] = __int_as_float(n_w); part[PW - 2] = __int_as_float(nb_w); }
            __syncthreads();
            if (G4R_DBGCLK(m)) t_h[3] = wall_clock64();
            if (wid == h) {
                for (int w = 0; w < SP_WAVES; ++w) {
#pragma unroll
                    for (int q = 0; q < MAXCH; ++q) {
                        const float4 x = *reinterpret_cast<const float4*>(sPart + w * PW + 4 * min(lane + 64 * q, nc4 - 1));
                        S[q].x += x.x; S[q].y += x.y; S[q].z += x.z; S[q].w += x.w;
                    }
                    Sb += sPart[w * PW + PW - 4];
                    n_e += __float_as_int(sPart[w * PW + PW - 3]);
                    nb_e += __float_as_int(sPart[w * PW + PW - 2]);
                }
            }
            __syncthreads();
            if (G4R_DBGCLK(m)) t_h[4] = wall_clock64();
        }
    }
    if (allsmp) {
        for (int cdup = 1; cdup < fl.z; ++cdup) {      // wave-uniform trip count
#pragma unroll
            for (int q = 0; q < MAXCH; ++q) { S[q].x += sk[q].x; S[q].y += sk[q].y; S[q].z += sk[q].z; S[q].w += sk[q].w; }
            Sb += bsk;
        }
        n_e = fl.z - 1; nb_e = fl.z - 1;
    }
    // ---- items with several occurrences: final rows from S + s_k, accumulator = the last occurrence's dA row
    if (owner && !single) {
        if constexpr (MAXCH == 2 && MOM) {
            // two chunks per lane with momentum: the pre-step velocity row is fetched AGAIN here rather than carried through the list
            // walk (nobody but this wave writes it) -- carried, it was the quad hipcc spilled right behind its load, with a vmcnt(0)
            // in front of the spill that every wave of the kernel paid for
#pragma unroll
            for (int q = 0; q < MAXCH; ++q) vz[q] = ld4(V + (size_t)item * W + 4 * min(lane + 64 * q, nc4 - 1));
        }
        finish(S, Sb, n_e + 1, nb_e + (bias ? 1 : 0));
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int c4 = lane + 64 * q;
            if (c4 < nc4) st4(A + (size_t)item * W + 4 * c4, ak[q]);
        }
        if (bias && lane == 0) m.accBy[item] = bak;
    }
    if (G4R_DBGCLK(m) && lane == 0 && k < R) {
        const long long t_end = wall_clock64();
        GAS long long* tr = G4R_DBGCLK(m) + 64 + 8 * k;
        tr[0] = t_start; tr[1] = t_own; tr[2] = t_col; tr[3] = t_app; tr[4] = t_end; tr[5] = (owner ? fl.z : 0) | ((t_h[0] ? t_h[0] - t_app : 0) << 20); tr[6] = load_ctx(st).t;
        tr[7] = (t_h[1] - t_h[0]) | ((t_h[2] - t_h[1]) << 16) | ((t_h[3] - t_h[2]) << 32) | ((t_h[4] - t_h[3]) << 48);
    }
}

template <int MAXCH, bool MOM>
__global__ __launch_bounds__(SP_WAVES * 64, MAXCH > 2 ? 2 : 4) void k_sparse_update(const DevModel* __restrict__ mp, StepState* st, int nblk_occ) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // workgroup 0 does the step bookkeeping (it depends on nothing the other workgroups produce; dispatched first, it is off the tail)
    sparse_update_block<MAXCH, MOM>(mp, st, nblk_occ, blockIdx.x == 0 ? nblk_occ : (int)blockIdx.x - 1, smem);
}

// ---------------------------------------------------------------------------------------------
// Deferred row updates: one flush launch per window of steps (a window = one replay of the step graph, <= 16 steps).
// A launch that moves one step's rows is too short for the HBM: the bare scatter pattern reaches 41-47 % of 8 TB/s at one step's rows and
// 72-77 % at 4-16 steps' rows (profiles/r02_micro_rows.json).  The plan (in_idx / out_idx of every step) and the sample store are known
// ahead, so for a window of steps it is known which occurrence is the LAST use of its item inside the window; if that occurrence also
// is the item's only one in its step, nothing gathers the row again before the window ends, and its update
//     P[item] -= step row (+ lr lmbd P[item]),   By[item] -= bias step            (gru4rec.py:420-431, one occurrence)
// can wait for the end of the window: same operands, same arithmetic, same bits as applying it at once (asserted: tests/
// test_gpu_defer.py).  The accumulators never wait (the gradient producers write them in place for single occurrences).
//   k_defer_scan pass 0: last_use[item] = max(global step) over the window's occurrences (X | Y | samples of every step)
//                pass 1: dcand[slot][k] = (last_use[item of occurrence k of step s] == that step)
//   k_update / k_sparse_update: a candidate that owns a single-occurrence item moves nothing and leaves dlist[slot][k] = item
//   k_sparse_flush: one wave per (step, occurrence) of the window: pending rows applied -- three row transfers each, in ONE launch over
//                up to 16 steps' rows; takes dcand / dlist back to 0 / -1.
// Windows never span a g4r_train_steps call, a sample-store refill or a compaction (the host loop launches scan, graph replay, flush).
__device__ __forceinline__ int defer_item(const DevModel& m, long long t, long long g, int k, int& table) {
    const int B = m.B, M = m.Mplan[t];
    table = 0;
    int item = -1;
    if (k < B) { if (k < M) item = m.in_idx[t * B + k]; table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? 0 : 1; }
    else if (k < 2 * B) { if (k - B < M) item = m.out_idx[t * B + (k - B)]; }
    else if (M > 0) item = m.ST[(size_t)(m.gl > 0 ? g % m.gl : 0) * m.ns + (k - 2 * B)];
    return item;
}
__global__ __launch_bounds__(256) void k_defer_scan(const DevModel* __restrict__ mp, long long t0, long long g0, int n, int pass) {
    const DevModel& m = *mp;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int R = m.R, s = (int)(idx / R), k = (int)(idx - (long long)s * R);
    if (s >= n) return;
    int table;
    const int item = defer_item(m, t0 + s, g0 + s, k, table);
    const size_t slot = G4R_SLOT(m, g0 + s) * (size_t)m.dRcap + k;
    if (item < 0) { if (pass) m.dcand[slot] = 0; return; }
    GAS int* lu = m.last_use + (size_t)table * m.n_items + item;
    if (pass == 0) atomicMax((int*)lu, (int)(g0 + s));
    else m.dcand[slot] = (*lu == (int)(g0 + s)) ? 1 : 0;
}
#define FL_NR 4      // pending entries per wave: their row requests are in flight together (rows of <= 256 floats)
__global__ __launch_bounds__(SP_WAVES * 64) void k_sparse_flush(const DevModel* __restrict__ mp, long long g0, int n) {
    const DevModel& m = *mp;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long e0 = ((long long)blockIdx.x * SP_WAVES + wid) * FL_NR;      // first entry of this wave (dRcap is a multiple of 8: a group stays inside one step)
    const int Rc = m.dRcap, s = (int)(e0 / Rc), k0 = (int)(e0 - (long long)s * Rc);
    if (s >= n) return;
    const size_t slot = G4R_SLOT(m, g0 + s) * (size_t)Rc + k0;
    int it[FL_NR];
#pragma unroll
    for (int u = 0; u < FL_NR; ++u) it[u] = m.dlist[slot + u];
    if (lane < FL_NR) { m.dcand[slot + lane] = 0; m.dlist[slot + lane] = -1; }
    const int B = m.B;
    const bool sep = m.embed_mode != G4R_EMBED_CONSTRAINED;
    const GAS float *sx = G4R_DSX(m, g0 + s), *sy = G4R_DSY(m, g0 + s), *sb = G4R_DSBY(m, g0 + s);
    const float lr = m.lr, lmbd = m.lmbd;
    // exactly sparse_update_block::finish for ONE occurrence (ss = 0 + s_k, fn = 1)
    auto upd = [&](float p0, float sl) { const float ss = 0.f + sl; const float reg = (lmbd > 0.f) ? lr * lmbd * p0 : 0.f; return p0 - ((lmbd > 0.f) ? ss + 1.0f * reg : ss); };
    int napp = 0, nbias = 0;
    const int wmax = max(m.Ein, m.Dtop);
    if (wmax <= 256) {
        float4 p[FL_NR], g[FL_NR];
        float bp = 0.f, bs = 0.f;
#pragma unroll
        for (int u = 0; u < FL_NR; ++u) {
            const int k = k0 + u, item = max(it[u], 0);
            const bool tE = k < B && sep;
            const int W = tE ? m.Ein : m.Dtop, cc = 4 * min(lane, (W >> 2) - 1);
            const GAS float* P = tE ? m.E : m.Wy;
            const GAS float* srow = (k < B) ? sx + (size_t)min(k, B - 1) * W : sy + (size_t)(k - B) * W;
            p[u] = ld4(P + (size_t)item * W + cc);
            g[u] = ld4(it[u] >= 0 ? srow + cc : P + (size_t)item * W + cc);
            if (lane == u && it[u] >= 0 && k >= B) { bp = m.By[item]; bs = sb[k - B]; }
        }
#pragma unroll
        for (int u = 0; u < FL_NR; ++u) {
            if (it[u] < 0) continue;      // wave-uniform
            const int k = k0 + u;
            const bool tE = k < B && sep;
            const int W = tE ? m.Ein : m.Dtop;
            GAS float* P = tE ? m.E : m.Wy;
            if (lane < (W >> 2)) st4(P + (size_t)it[u] * W + 4 * lane, make_float4(upd(p[u].x, g[u].x), upd(p[u].y, g[u].y), upd(p[u].z, g[u].z), upd(p[u].w, g[u].w)));
            if (lane == u && k >= B) m.By[it[u]] = upd(bp, bs);
            ++napp; nbias += (k >= B) ? 1 : 0;
        }
    } else {
        for (int u = 0; u < FL_NR; ++u) {
            if (it[u] < 0) continue;
            const int k = k0 + u, item = it[u];
            const bool tE = k < B && sep;
            const int W = tE ? m.Ein : m.Dtop, nc4 = W >> 2;
            GAS float* P = tE ? m.E : m.Wy;
            const GAS float* srow = (k < B) ? sx + (size_t)k * W : sy + (size_t)(k - B) * W;
            float4 p[4], g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // rows of <= 1024 floats: up to four quads per lane, all requested together
                const int cc = 4 * min(lane + 64 * q, nc4 - 1);
                p[q] = ld4(P + (size_t)item * W + cc);
                g[q] = ld4(srow + cc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane + 64 * q < nc4)
                    st4(P + (size_t)item * W + 4 * (lane + 64 * q), make_float4(upd(p[q].x, g[q].x), upd(p[q].y, g[q].y), upd(p[q].z, g[q].z), upd(p[q].w, g[q].w)));
            if (k >= B && lane == 0) m.By[item] = upd(m.By[item], sb[k - B]);
            ++napp; nbias += (k >= B) ? 1 : 0;
        }
    }
    // statistics (bench.py, tests): 1024 counter pairs, one per workgroup id mod 1024 -- a single counter serialised 10^5 atomics per launch
    // (11-13 ns each: the launch took milliseconds)
    if (lane == 0 && napp) { GAS unsigned* ds = m.dstat + 2 * (blockIdx.x & 1023u); atomicAdd((unsigned*)ds, (unsigned)napp); if (nbias) atomicAdd((unsigned*)ds + 1, (unsigned)nbias); }
}

// Single GPU: the dense-gradient tiles (+ fused dense Adagrad) and the sparse row update are independent of each other
// (the tiles read layer-0 input rows from yin0, not from the table), so they share ONE launch: blocks [0, ntiles) are
// dense tiles, the rest sparse-update blocks.  One dispatch (~4.5 us) less per step.
static_assert(GT_NTH_FEW == SP_WAVES * 64, "both roles use the same workgroup size");
template <int MAXCH, int DT, bool MOM>
__global__ __launch_bounds__(SP_WAVES * 64, 4) void k_update(const DevModel* __restrict__ mp, StepState* st, const DenseTile* __restrict__ tiles_,
                                                             int ntiles, int nblk_occ) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // workgroup 0: step bookkeeping (dispatched first: off the tail), then the dense tiles, then the sparse-update workgroups
    // (interleaving the two kinds in dispatch order was measured: no change -- both draw on L2 / fabric bandwidth)
    const int b = (int)blockIdx.x - 1;
    if (b < 0) sparse_update_block<MAXCH, MOM>(mp, st, nblk_occ, nblk_occ, smem);
    else if (b < ntiles) {
        const int t = G4R_XCD_TILE(b, ntiles);      // neighbouring tiles of the table share their X rows: keep them on one XCD
        dense_grad_tile<DT>(*mp, st, tiles_, t, smem);
    } else sparse_update_block<MAXCH, MOM>(mp, st, nblk_occ, b - ntiles, smem);
}

// ---------------------------------------------------------------------------------------------
// Exact-replica mode (g4r_config::sparse_exact): the (last, first, count) table of the CONCATENATED occurrence list.  The forward
// kernels do not publish their own occurrences in this mode (`xmode != 0`); k_exact_occ, behind the all-gather in stream order,
// publishes all xn * R occurrences with their global positions, and the owners in k_sparse_update_generic take the entries back to zero.
// Position K of the exchanged occurrence list -> (rank block q, occurrence k of that rank's X | Y | samples list).
//   xmode 1 / 2 (SUM / MEAN forms): the ranks' lists one behind the other, K = q * R + k, xn * R entries;
//   xmode 3 (REDUCE form; all ranks draw the SAME negatives): X | Y of rank 0, X | Y of rank 1, ..., then the sample part ONCE
//     (xn * 2B + ns entries): a sample entry stands for that column of every rank, its gradient row is the sum over the ranks
//     (block -1 below) -- the all-reduce of the negatives' gradient rows a data-parallel step owes the reference's shared row of
//     negatives (gru4rec.py:436-437); the list then is exactly the occurrence list of ONE batch of xn * B rows.
struct XPos { int q, k; };
__device__ __forceinline__ int xlist_len(const DevModel& m) { return m.xmode == 3 ? m.xn * 2 * m.B + m.ns : m.xn * m.R; }
__device__ __forceinline__ XPos xlist_pos(const DevModel& m, int K) {
    XPos p;
    if (m.xmode == 3) {
        const int nxy = m.xn * 2 * m.B;
        if (K >= nxy) { p.q = -1; p.k = 2 * m.B + (K - nxy); }
        else { p.q = K / (2 * m.B); p.k = K - p.q * 2 * m.B; }
    } else { p.q = K / m.R; p.k = K - p.q * m.R; }
    return p;
}
// Item of an entry of the exchanged list.  A shared negative of the REDUCE form (q < 0) stands for one score column of EVERY rank: its
// id is taken from the first block that holds it (a rank in the padded tail of its plan -- M = 0 -- stages -1 for its sample
// columns while the other ranks still train; round 4 read rank 0's block only and dropped every rank's update of the negatives then).
__device__ __forceinline__ int xlist_item(const DevModel& m, XPos p) {
    if (p.q >= 0) return ((const GAS int*)(m.xbase + (long long)p.q * m.xstride))[p.k];
    int item = -1;
    for (int r = 0; r < m.xn; ++r) {
        const int v = ((const GAS int*)(m.xbase + (long long)r * m.xstride))[p.k];
        if (item < 0) item = v;
    }
    return item;
}
__global__ __launch_bounds__(256) void k_exact_occ(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    const int K = blockIdx.x * 256 + threadIdx.x, R = xlist_len(m);
    if (K >= R) return;
    const XPos ps = xlist_pos(m, K);
    const int k = ps.k;
    const int item = xlist_item(m, ps);
    if (ps.q < 0) {
        // the ranks must have drawn the SAME negatives (one sample stream: GRU4Rec._create_model seeds every rank alike in this mode;
        // a C-API caller may not): their gradient rows are summed under ONE id.  A mismatch poisons the step's cost (NaN: the
        // host's NaN check stops the run, gru4rec.py:626) instead of silently training items under other items' ids.
        for (int r = 0; r < m.xn; ++r) {
            const int v = ((const GAS int*)(m.xbase + (long long)r * m.xstride))[k];
            if (v >= 0 && v != item) m.st->nan_flag = 2;
        }
    }
    if (item < 0) return;
    const bool tableE = (k < m.B && m.embed_mode != G4R_EMBED_CONSTRAINED);
    int* fl = (int*)m.occ_fl + 4 * ((tableE ? (size_t)m.n_items : 0) + item);
    atomicMax(fl, K + 1);
    atomicMax(fl + 1, R - K);
    atomicAdd(fl + 2, 1);
}

// ---------------------------------------------------------------------------------------------
// Sparse update of the generic optimizer path (rmsprop / adadelta / adam / plain SGD, and adagrad under grad_cap).
// Same ownership scheme as k_sparse_update (the wave of an item's last occurrence owns its rows; first / last / count table),
// but the gradient rows are RAW: the owner sums S = sum g, Q = sum g^2 (and adagrad's per-occurrence scaled sum) over all
// occurrences of the item, applies opt_rule once per element and writes parameter, statistics and velocity.  With the
// reference's "accurate" duplicate handling (gru4rec.py:321-326,349-358,373-378) every occurrence of an item sees the same
// final statistic, so sums are all that is needed.  Simple rather than fast: the owner walks its occurrences alone.
// Exact-replica mode of N > 1 (g4r_config::sparse_exact): the occurrence list is the concatenation of the xn ranks' lists, K = q * R + k
// (block q of the exchange buffer, occurrence k of that rank: X | Y | samples), and the gradient rows are read from the owning
// rank's block; the duplicate semantics -- per-occurrence Adagrad scaling with the pre-step accumulator, increments accumulate,
// statistics / velocity take the LAST occurrence -- hold over the concatenated list, i.e. ranks count as later occurrences in rank
// order.  xn == 1 is the single-rank generic path.
template <int MAXCH>
__global__ __launch_bounds__(SP_WAVES * 64, MAXCH == 1 ? 4 : 2) void k_sparse_update_generic(const DevModel* __restrict__ mp, StepState* st, int nblk_occ, int nda) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int B = m.B, Rl = m.R, xn = m.xn, R = xlist_len(m);      // Rl: occurrences per rank, R: of the whole (exchanged) list
    // grid: [0, nda) the dense rule on the (all-reduced) flat gradient -- it shares the launch, independent of the item rows, and is
    // dispatched first --, then nblk_occ row blocks, then the bookkeeping block
    if ((int)blockIdx.x < nda) {
        const int i = (int)blockIdx.x * (SP_WAVES * 64) + tid;
        if (i < m.dense_count) dense_apply_elem(m, i);
        return;
    }
    const int bid = (int)blockIdx.x - nda;
    if (bid == nblk_occ) {       // bookkeeping block, as in k_sparse_update
        const StepCtx c = load_ctx(st);
        const int Mn = m.Mplan[c.t + 1];
        if (wid == 0) {
            float s = 0.f;
            for (int i = lane; i < c.M; i += 64) s += m.lossrow[i];
            s = wave_sum(s);
            if (lane == 0) {
                GAS StepState* sg = (GAS StepState*)st;
                const float cost = (sg->nan_flag == 2) ? __builtin_nanf("") : s * m.inv_B;      // 2: the ranks' negatives differ (k_exact_occ)
                m.loss_steps[c.t] = cost;
                if (isnan(cost) && sg->nan_flag == 0) sg->nan_flag = 1;
                sg->t_a = c.t + 1;
                sg->g_a = c.g + 1;
                sg->M_a = Mn;
            }
        }
        stage_step_inputs(m, c.t + 1, c.g + 1, Mn, tid, SP_WAVES * 64);
        return;
    }
    const GAS float* xb = m.xbase;
    const long long xs = m.xstride;
    const int Rpad = ((R + 255) & ~255) + 256;
    int* sOcc = reinterpret_cast<int*>(smem);
    int* myList = sOcc + Rpad + 64 * wid;
    // Occurrences are strided over the workgroups (wave w of workgroup b takes k = w * nblk + b), as in k_sparse_update: the owners
    // of the popular items -- last occurrences, at the end of the list -- do not share a few workgroups.  The item of k comes straight
    // from the exchanged list in memory; the list is staged in LDS only by workgroups in which some wave owns a REPEATED item (one
    // barrier-or), so the common wave -- owner of a single occurrence -- makes two round trips (item; entry + rows) and stores.
    const int k = wid * nblk_occ + bid;
    const XPos pk = xlist_pos(m, min(k, R - 1));
    const int item = k < R ? xlist_item(m, pk) : -1;
    const int kl = pk.k;                          // local occurrence of k (position in its rank's X | Y | samples list)
    auto is_x = [&](int j) { return xlist_pos(m, j).k < B; };      // an input occurrence (table E when the tables are separate; no output bias)
    const float xscale = (m.xmode == 3) ? G4R_MUT_XSCALE(1.0f / (float)xn) : 1.0f;      // REDUCE form: gradients of the GLOBAL batch (cost / (xn * B))
    const bool constrained = (m.embed_mode == G4R_EMBED_CONSTRAINED);
    const bool tableE = (kl < B && !constrained);
    GAS int* flp = m.occ_fl + 4 * ((tableE ? (size_t)m.n_items : 0) + max(item, 0));
    const int4 fl = ldi4(flp);
    // occurrence range sharing a table with k: constrained -> everything; separate tables -> the X parts (table E) or the
    // Y | samples parts (table Wy) of all blocks: `same_table(j)` filters the scan below
    const int lo = (constrained || kl < B || xn > 1) ? 0 : B;
    const int first_j = max(lo, R - fl.y);
    auto same_table = [&](int j) { return constrained || is_x(j) == tableE; };
    GAS float *P = tableE ? m.E : m.Wy, *A = tableE ? m.accE : m.accWy, *A2 = tableE ? m.acc2E : m.acc2Wy,
              *Cn = tableE ? m.cntE : m.cntWy, *V = tableE ? m.velE : m.velWy;
    const int W = tableE ? m.Ein : m.Dtop, nc4 = W >> 2;
    // Narrow rows (one quad per lane, nc4 <= 32): a row needs only LW = 16 / 32 lanes, so every load instruction of the repeated-item
    // walk fetches RPI = 64 / LW occurrences, lane group `sub` taking occurrence i0 + u * RPI + sub; the groups' partial sums are
    // combined with lane shuffles.  All row accesses use the lane's column `col`; the result is written by group 0 (col == lane).
    const int LW = (MAXCH == 1) ? (nc4 <= 16 ? 16 : (nc4 <= 32 ? 32 : 64)) : 64;
    const int RPI = 64 / LW, sub = lane / LW, col = lane & (LW - 1);
    // output bias (gru4rec.py:486-489: By is indexed by Y | samples only).  In one rank's list the X occurrences come first, so an
    // item whose LAST occurrence is an input has no bias occurrence at all; in a concatenated list (xn > 1) a later rank's input may
    // follow an earlier rank's target / negative: the owner then still updates the bias, from the bias occurrences the scan finds,
    // and "the last occurrence" of the bias statistics is the last of THOSE
    const bool bias_own = (kl >= B), bias_maybe = bias_own || (xn > 1 && fl.z > 1 && constrained), mom = m.mom > 0.f;
    const int adapt = m.adapt;
    const float v1 = m.ap0, v3 = m.ap1, lr = m.lr, lmbd = m.lmbd, momc = m.mom, clip = m.gclip[0];
    const bool adagrad = (adapt == G4R_ADAPT_ADAGRAD);
    const int oSx = m.xoffSx, oSy = m.xoffSy, oSB = m.xoffSBy;
    // row state.  LATE (rows of four quads per lane): what only the final rule reads -- parameter, second statistic, count, velocity --
    // is requested behind the walk over the occurrences instead of in front of it: 64 registers less held across the walk (the
    // variant had 85 spilled registers; these rows pay one more round trip, once per owned item)
    constexpr bool LATE = MAXCH >= 4;
    float4 p0[MAXCH], a0[MAXCH], u0[MAXCH], c0[MAXCH], w0[MAXCH], S[MAXCH], Q[MAXCH], T1[MAXCH], gk[MAXCH];
    auto sq = [](float4 x) { return make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w); };
    auto add4 = [](float4& a, float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    auto ada = [](float4 g, float4 a) {      // g / sqrt(a + g^2 + eps), per component (v_rsq_f32, ~1 ulp, as in the Adagrad producers)
        return make_float4(g.x * frsq(a.x + g.x * g.x + G4R_EPS_ADAGRAD), g.y * frsq(a.y + g.y * g.y + G4R_EPS_ADAGRAD),
                           g.z * frsq(a.z + g.z * g.z + G4R_EPS_ADAGRAD), g.w * frsq(a.w + g.w * g.w + G4R_EPS_ADAGRAD));
    };
    const float gsc = clip * xscale;
    auto grow = [&](int j, int q) {          // clipped gradient row chunk of occurrence j (of the exchanged list)
        const XPos pj = xlist_pos(m, j);
        const int jl = pj.k, c = 4 * min(col + 64 * q, nc4 - 1);
        const size_t ro = (jl < B) ? (size_t)oSx + (size_t)jl * W : (size_t)oSy + (size_t)(jl - B) * W;
        float4 g = ld4(xb + (long long)max(pj.q, 0) * xs + ro + c);
        if (pj.q < 0)                          // REDUCE form, a shared negative: the sum over the ranks' rows of this column, in rank order
            for (int r2 = 1; r2 < xn; ++r2) { const float4 h = ld4(xb + (long long)r2 * xs + ro + c); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        return make_float4(gsc * g.x, gsc * g.y, gsc * g.z, gsc * g.w);
    };
    auto bgrad = [&](int j) {                // clipped output-bias gradient of occurrence j (only for j among Y | samples of its block)
        const XPos pj = xlist_pos(m, j);
        const int o = oSB + max(pj.k - B, 0);
        float g = (xb + (long long)max(pj.q, 0) * xs)[o];
        if (pj.q < 0)
            for (int r2 = 1; r2 < xn; ++r2) g += (xb + (long long)r2 * xs)[o];
        return gsc * g;
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const size_t o = (size_t)max(item, 0) * W + 4 * min(col + 64 * q, nc4 - 1);      // (a wave without an occurrence reads row 0 and drops out below)
        a0[q] = ld4(A + o);
        if constexpr (!LATE) { p0[q] = ld4(P + o); u0[q] = A2 ? ld4(A2 + o) : z4; c0[q] = Cn ? ld4(Cn + o) : z4; w0[q] = mom ? ld4(V + o) : z4; }
        gk[q] = grow(min(k, R - 1), q);
        S[q] = z4; Q[q] = z4; T1[q] = z4;
    }
    float bp0 = 0.f, ba0 = 0.f, bu0 = 0.f, bc0 = 0.f, bw0 = 0.f, bgk = 0.f, bS = 0.f, bQ = 0.f, bT1 = 0.f;
    int lastb = bias_own ? k : -1;      // last bias occurrence of the item found so far
    if (bias_own || (xn > 1 && constrained)) {      // (a superset of bias_maybe that does not wait for the entry)
        const int it0 = max(item, 0);
        bp0 = m.By[it0]; ba0 = m.accBy[it0]; bgk = bias_own ? bgrad(min(k, R - 1)) : 0.f;
        if (m.acc2By) bu0 = m.acc2By[it0];
        if (m.cntBy) bc0 = m.cntBy[it0];
        if (mom) bw0 = m.velBy[it0];
    }
    // (the row state, the gradient row of k and the bias state above are in flight: requested together with the entry)
    const bool owner = item >= 0 && fl.x == k + 1;      // the last occurrence of the item
    if (__syncthreads_or((owner && fl.z > 1) ? 1 : 0)) {
        for (int j = tid; j < Rpad; j += SP_WAVES * 64) {
            const XPos pj = xlist_pos(m, min(j, R - 1));
            sOcc[j] = j < R ? xlist_item(m, pj) : -2;
        }
        __syncthreads();
    }
    if (!owner) return;
    if (lane == 0) {
        *(GAS int4*)flp = make_int4(0, 0, 0, 0);
        if (m.touched) m.touched[(tableE ? (size_t)m.n_items : 0) + item] = 1;
    }
    // earlier occurrences in [first, k), 64 per pass, NB rows per round trip
    int n = 1, nb = bias_own ? 1 : 0;
    // MEAN form of the exact-replica mode (sparse_exact = 2; what the GPU-local mode's reconciliation does, taken every step): the
    // item's parameter increment is the MEAN over the ranks that touch it of each rank's own increment (N full-size Adagrad steps
    // from one starting point must not add up: measured), and the Adagrad accumulator takes the SUM over those
    // ranks of each rank's last-occurrence increment.  nq / nqb: touching ranks of the row / of the bias; Aadd / bAadd: the
    // accumulator increments of the ranks' last occurrences.  (An item with more than 64 earlier occurrences: rank boundaries that
    // fall on a pass boundary are not seen -- a deterministic approximation, identical on every rank.)
    const bool xmean = m.xmode == 2;
    int nq = 1, nqb = bias_own ? 1 : 0;
    float4 Aadd[MAXCH];
    float bAadd = 0.f;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) Aadd[q] = z4;
    if (fl.z > 1) {
        for (int pass = 0;; ++pass) {
            int idx = 0;
            // 256 list entries per step (one 16-byte LDS read per lane; reads past the list stay inside Rpad and never match); the
            // table filter -- an integer division per entry -- only where the tables are separate
            for (int base = first_j & ~255; base < k; base += 256) {
                const int4 v = *reinterpret_cast<const int4*>(sOcc + base + 4 * lane);
                const int j = base + 4 * lane;
                bool hh[4] = {v.x == item && j >= first_j && j < k, v.y == item && j + 1 >= first_j && j + 1 < k,
                              v.z == item && j + 2 >= first_j && j + 2 < k, v.w == item && j + 3 >= first_j && j + 3 < k};
                if (__ballot(hh[0] || hh[1] || hh[2] || hh[3]) == 0) continue;
                if (!constrained) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hh[e] = hh[e] && same_table(j + e);
                }
                // ascending occurrence order = lane-major: all matches of lower lanes first, then this lane's earlier entries
                int below = 0, total = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned long long mk = __ballot(hh[e]);
                    below += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    total += __popcll(mk);
                }
                int ord = idx - 64 * pass + below;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (hh[e]) {
                        if (ord >= 0 && ord < 64) myList[ord] = j + e;
                        ++ord;
                    }
                }
                idx += total;
            }
            const int cnt = min(idx - 64 * pass, 64);
            const int myj = lane < cnt ? myList[lane] : -1;
            const bool final_pass = idx <= 64 * (pass + 1);
            // the last occurrence of each rank among the hits: its successor (the next hit, or k behind the final pass) is another rank's
            int lastrow = 0;
            if (xmean) {
                const int nxt_l = myList[min(lane + 1, 63)];
                const int nxt = (lane + 1 < cnt) ? nxt_l : (final_pass ? k : myj);
                lastrow = (lane < cnt && (myj / Rl) != (nxt / Rl)) ? 1 : 0;
                nq += __popcll(__ballot(lastrow != 0));
            }
            if (bias_maybe) {
                const bool isb = myj >= 0 && !is_x(myj);
                const float g = isb ? bgrad(myj) : 0.f;
                bS += wave_sum(g); bQ += wave_sum(g * g);
                bT1 += wave_sum(isb ? g / sqrtf(ba0 + g * g + G4R_EPS_ADAGRAD) : 0.f);
                nb += __popcll(__ballot(isb));
                if (!bias_own) lastb = max(lastb, (int)wave_max(isb ? (float)myj : -1.f));      // (list positions < 2^24: exact as floats)
                if (xmean) {
                    // the next BIAS hit behind this lane (or k, if the owner is a bias occurrence itself)
                    const unsigned long long mb = __ballot(isb);
                    const unsigned long long hi = (lane < 63) ? (mb >> (lane + 1)) : 0ull;
                    const int nl = hi ? lane + 1 + (int)__builtin_ctzll(hi) : -1;
                    const int nbj_l = myList[max(nl, 0) & 63];
                    const int nbj = nl >= 0 ? nbj_l : ((final_pass && bias_own) ? k : -1);
                    const bool lastbias = isb && (nbj < 0 || (myj / Rl) != (nbj / Rl));
                    nqb += __popcll(__ballot(lastbias));
                    bAadd += wave_sum(lastbias ? g * g : 0.f);
                }
            }
            // NB gradient rows per round trip: the sampler repeats the head of the catalogue 20-50 x per step, and the owner walks
            // its occurrences alone -- with 4 rows per trip the hottest item's 13 dependent trips set the launch's length
            constexpr int NB = LATE ? 2 : 4;      // (16 rows per trip at MAXCH = 1 cost 40 registers -> one workgroup per CU instead of two: the launch got slower)
            for (int i0 = 0; i0 < cnt; i0 += NB * RPI) {
                float4 g[NB][MAXCH];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int jj = __shfl(myj, min(i0 + u * RPI + sub, cnt - 1) & 63);
#pragma unroll
                    for (int q = 0; q < MAXCH; ++q) g[u][q] = grow(jj, q);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int ri = i0 + u * RPI + sub;
                    const bool lr_u = __shfl(lastrow, min(ri, cnt - 1) & 63) != 0;
                    if (ri < cnt) {
#pragma unroll
                        for (int q = 0; q < MAXCH; ++q) {
                            add4(S[q], g[u][q]); add4(Q[q], sq(g[u][q]));
                            if (adagrad) add4(T1[q], ada(g[u][q], a0[q]));
                            if (lr_u) add4(Aadd[q], sq(g[u][q]));
                        }
                    }
                }
            }
            n += cnt;
            if (idx <= 64 * (pass + 1)) break;
        }
    }
    if constexpr (MAXCH == 1) {
        if (RPI > 1 && fl.z > 1) {      // the lane groups' partial sums -> every lane (group 0 writes the row)
            auto comb = [&](float4& v) {
                for (int off = LW; off < 64; off <<= 1) {
                    v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
                }
            };
            comb(S[0]); comb(Q[0]); comb(T1[0]); comb(Aadd[0]);
        }
    }
    if constexpr (LATE) {
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const size_t o = (size_t)item * W + 4 * min(col + 64 * q, nc4 - 1);
            p0[q] = ld4(P + o); u0[q] = A2 ? ld4(A2 + o) : z4; c0[q] = Cn ? ld4(Cn + o) : z4; w0[q] = mom ? ld4(V + o) : z4;
        }
    }
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        add4(S[q], gk[q]); add4(Q[q], sq(gk[q]));
        if (adagrad) add4(T1[q], ada(gk[q], a0[q]));
        const float fn = (float)n;
        const float pp[4] = {p0[q].x, p0[q].y, p0[q].z, p0[q].w}, aa[4] = {a0[q].x, a0[q].y, a0[q].z, a0[q].w};
        const float uu[4] = {u0[q].x, u0[q].y, u0[q].z, u0[q].w}, cc[4] = {c0[q].x, c0[q].y, c0[q].z, c0[q].w};
        const float ww[4] = {w0[q].x, w0[q].y, w0[q].z, w0[q].w}, ss[4] = {S[q].x, S[q].y, S[q].z, S[q].w};
        const float qq[4] = {Q[q].x, Q[q].y, Q[q].z, Q[q].w}, tt[4] = {T1[q].x, T1[q].y, T1[q].z, T1[q].w};
        const float gg[4] = {gk[q].x, gk[q].y, gk[q].z, gk[q].w};
        float pn[4], an[4], un[4], cn[4], vn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const OptOut o = opt_rule(adapt, v1, v3, false, aa[e], uu[e], cc[e], ss[e], qq[e], tt[e], gg[e], fn);
            const float reg = (lmbd > 0.f) ? lmbd * pp[e] : 0.f;
            const float dsum = lr * (o.G + fn * reg);                  // sum of the per-occurrence deltas (gru4rec.py:419-423)
            const float ad[4] = {Aadd[q].x, Aadd[q].y, Aadd[q].z, Aadd[q].w};
            an[e] = (xmean && adagrad) ? o.A + ad[e] : o.A; un[e] = o.U; cn[e] = o.C;
            const float inc = mom ? (fn * (momc * ww[e]) - dsum) : -dsum;      // the parameter increment of all occurrences together
            vn[e] = mom ? momc * ww[e] - lr * (o.gl + reg) : 0.f;
            pn[e] = pp[e] + (xmean ? G4R_MUT_XSCALE(inc / (float)nq) : inc);
        }
        const int c4 = lane + 64 * q;
        if (c4 < nc4) {
            const size_t o = (size_t)item * W + 4 * c4;
            st4(P + o, make_float4(pn[0], pn[1], pn[2], pn[3]));
            st4(A + o, make_float4(an[0], an[1], an[2], an[3]));
            if (A2) st4(A2 + o, make_float4(un[0], un[1], un[2], un[3]));
            if (Cn) st4(Cn + o, make_float4(cn[0], cn[1], cn[2], cn[3]));
            if (mom) st4(V + o, make_float4(vn[0], vn[1], vn[2], vn[3]));
        }
    }
    if (bias_maybe && nb > 0 && lane == 0) {
        if (bias_own) { bS += bgk; bQ += bgk * bgk; bT1 += bgk / sqrtf(ba0 + bgk * bgk + G4R_EPS_ADAGRAD); }
        else bgk = bgrad(lastb);      // the sums already hold every bias occurrence; statistics / velocity follow the last of them
        const float fb = (float)nb;
        const OptOut o = opt_rule(adapt, v1, v3, false, ba0, bu0, bc0, bS, bQ, bT1, bgk, fb);
        const float reg = (lmbd > 0.f) ? lmbd * bp0 : 0.f;
        const float dsum = lr * (o.G + fb * reg);
        // MEAN form: pre-step value + the ranks' last-occurrence increments.  bAadd holds those of the bias hits of the scan; the owner's
        // own (o.A - ba0) joins them only when the owner IS a bias occurrence -- otherwise the last bias hit is already in bAadd (round 4
        // added it twice: found by the oracle-as-replicas test of this form)
        m.accBy[item] = (xmean && adagrad) ? (bias_own ? o.A + bAadd : ba0 + bAadd) : o.A;
        if (m.acc2By) m.acc2By[item] = o.U;
        if (m.cntBy) m.cntBy[item] = o.C;
        const float inc = mom ? (fb * (momc * bw0) - dsum) : -dsum;
        if (mom) m.velBy[item] = momc * bw0 - lr * (o.gl + reg);
        m.By[item] = bp0 + (xmean ? G4R_MUT_XSCALE(inc / (float)max(nqb, 1)) : inc);
    }
}

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
