// g4r_update_kernels.cuh -- part of g4r_step_kernels.cuh (included there, in order; needs its prelude).  Holds gradients of the dense weights and every update rule: k_onehot_step, k_dense_grad, k_grad_sqsum / k_grad_clip, k_dense_apply, k_sparse_update, k_defer_scan / k_sparse_flush, k_update, k_exact_occ, k_sparse_update_generic.
#pragma once
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
