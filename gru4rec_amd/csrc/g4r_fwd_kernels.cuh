// g4r_fwd_kernels.cuh -- part of g4r_step_kernels.cuh (included there, in order; needs its prelude).  Holds forward of the step: k_gru_p1 / k_gru_p2 (also the prediction path), k_gru_fwd_fused, k_score_fwd.
#pragma once
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
