// GRU kernels for WIDE layers (D a multiple of 64, >= 256 units; gfx950) on 64 x 64 tiles of v_mfma_f32_32x32x2_f32 (g4r_gemm.cuh) with the
// K range of a tile cut into slices, one workgroup each, and the slices left as PARTIAL SUMS that the kernel which consumes the
// product anyway adds up behind the kernel boundary:
//   k_gru_p1s   V = [y | H] [Wx ; 0 | Wrz] as partial sums vp[slice][B][3D]      -> k_gru_gate adds them, bias, r / Hr / z
//   k_gru_bwd_bw dy = dV Wx^T as partial sums dyp[slice][B][IN]                   -> the lower layer's k_gru_bwd_pre, or (layer 0) k_finish_rows
//   k_dense_grad2 the dense gradients on 64 x 64 tiles (contraction over the batch: nothing to slice) as a launch of its own, with the
//                 row-finishing workgroups of layer 0 behind its tiles
//
// Why (BASELINE configs[2]: B = 240, D = 512; profiles/r04_*, profiles/r05_experiments.md): a wide layer's products are a few dozen
// 64 x 64 tiles -- the round-1 kernels keep the whole K range in one workgroup (k_gru_bwd_b: 128 workgroups of 393 KB of operands,
// 17.5 us for 377 MFLOP), and a tile's K loop runs at the MFMA rate of ONE CU (0.7 us per 32-deep stage).  Slicing K gives every CU a
// share; what it must not do is JOIN the slices inside the launch: the round-3 / round-4 reviews asked for a last-arriver join
// (write-through partials, a ticket per tile, the last workgroup adds them up); it was built for all five products of the layer, was
// bit-exact and deterministic, and LOST to the round-1 kernels everywhere (k_gru_p1 28.6 vs 24.7 us, k_gru_p2 13.5 vs 9.9) because a
// join costs ~7 us of publish + ticket + re-read, more than a slice's K loop (commit 163e0f5..410f7c0 have that code; the log has the
// numbers).  A kernel boundary that exists anyway is the cheaper hand-over: dy's slices cost their consumer 8-12 extra loads per
// element.  k_gru_bwd_b 17.5 -> 7.0 us (configs[2]), 10.9 -> 5.6 us (configs[3] shape, where it needs k_dense_grad2 as its consumer's
// launch and the merged k_update is the better update: not the default there).
// The reference's math is unchanged (gru4rec.py:471-479 and its T.grad, :383-384); only the fp32 summation order differs (k-ordered
// inside a slice, slices added in slice order: deterministic, graph replay == eager).
#pragma once
#include "g4r_step_kernels.cuh"

// work item -> (column tile, K slice, row tile): row tiles innermost, so that the workgroups of one XCD (G4R_XCD_TILE) that share a
// weight slab (same columns, same K slice) sit next to each other
struct WideItem { int ct, s, rt; };
__device__ __forceinline__ WideItem wide_item(int idx, int nsplit, int nrt) {
    WideItem w;
    const int per = nsplit * nrt;
    w.ct = idx / per;
    const int rem = idx - w.ct * per;
    w.s = rem / nrt;
    w.rt = rem - w.s * nrt;
    return w;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GRU phase 1 (training) as K-slice partial sums, gru4rec.py:472:  V[B, 3D] = [y | H] [Wx ; 0 | Wrz].  K slices: `ny` slices of `kys`
// (<= 128) input units (A = the layer's input rows: gathered table rows with embedding dropout for layer 0), then `nh` slices of
// `khs` hidden units (A = H; only for the r / z columns: the candidate columns have no hidden part here -- theirs is (H r) Wh, phase
// 2).  B = rows of Wx / Wrz ([k][n], K-major).  A bare GEMM: the whole slice of both operands is requested at once
// (gemm_tile2k_full), the partial tile goes to vp[slice][B][3D]; 98-103 registers, four workgroups per CU.
// Embedding dropout: the Philox masks of a thread's staging slots (one quad per 16-deep chunk: row tid >> 2, k offset 4 (tid & 3))
// are drawn up front, next to the operand requests, as 4 bits per chunk.  Column tile 0 publishes the masked layer-0 input rows
// (yin0) for the dense-gradient tiles and the X part of the step's occurrence list, as k_gru_p1 does.
__global__ __launch_bounds__(256, 4) void k_gru_p1s(const DevModel* __restrict__ mp, StepState* st, int l, int first, int ny, int nh, int kys, int khs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int tid = threadIdx.x;
    const StepCtx c = first ? load_ctx_first(st) : load_ctx(st);
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D, B = m.B, M = c.M;
    const unsigned g = (unsigned)c.g;
    const int nrt = (B + 63) >> 6, nct_c = D >> 6;
    // candidate column tiles come first (ny slices each), then the r / z column tiles (ny + nh slices each)
    const int base_c = nct_c * ny * nrt;
    const int idx = G4R_XCD_TILE(blockIdx.x, gridDim.x);
    const bool cand = idx < base_c;
    WideItem w = wide_item(cand ? idx : idx - base_c, cand ? ny : ny + nh, nrt);
    if (!cand) w.ct += nct_c;
    const int m0 = w.rt * 64, n0 = w.ct * 64, s = w.s;
    const GAS int* gidx = m.cur_in;      // staged by the previous step's bookkeeping
    if (l == 0 && w.ct == 0 && s == 0 && tid < 64) {
        // the X part of the step's occurrence list (k_sparse_update), also for row tiles past the active batch
        const int row = m0 + tid;
        if (row < B) {
            const int item = row < M ? gidx[row] : -1;
            m.occ_idx[row] = item;
            if (item >= 0 && m.xmode == 0) {
                int* fl = (int*)m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED ? 0 : (size_t)m.n_items) + item);
                atomicMax(fl, row + 1);
                atomicMax(fl + 1, m.R - row);
                atomicAdd(fl + 2, 1);
            }
        }
    }
    if (m0 >= M) return;
    const bool ysl = s < ny;
    const int ks = ysl ? s * kys : (s - ny) * khs;
    const int Klen = ysl ? min(kys, IN - ks) : min(khs, D - ks);
    const GAS float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const GAS float* ysrc = l > 0 ? m.hd[l - 1] : nullptr;
    const GAS float* Hcur = m.H[l][g & 1];
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wrz = m.dense_p + m.offWrz[l];
    auto aprov = [&](int r) -> const GAS float* {
        const int row = m0 + r;
        if (row >= M) return nullptr;
        if (!ysl) return Hcur + (size_t)row * D + ks;
        if (l == 0) return table + (size_t)gidx[row] * IN + ks;
        return ysrc + (size_t)row * IN + ks;
    };
    auto bprov = [&](int kk, int kr, int kc) -> const GAS float* {
        const int k = kk + kr;
        if (k >= Klen) return nullptr;
        return ysl ? Wx + (size_t)(ks + k) * D3 + n0 + kc : Wrz + (size_t)(ks + k) * (2 * D) + (n0 - D) + kc;
    };
    // dropout bits of this thread's staging slots, 4 per chunk (<= 8 chunks)
    const int sr = tid >> 2, sc = 4 * (tid & 3);
    const bool dropping = ysl && l == 0 && m.drop_e > 0.f;
    const float retain = 1.0f - m.drop_e, inv_retain = 1.0f / retain;
    unsigned dm = ~0u;
    if (dropping) {
        dm = 0u;
        const unsigned s0 = (unsigned)m.seed, s1 = (unsigned)(m.seed >> 32);
        const int nch = Klen >> 4;
        for (int i = 0; i < nch; ++i) {
            const Philox4 p = philox4x32_10((unsigned)((ks + 16 * i + sc) >> 2), (unsigned)(m0 + sr), g, G4R_STREAM_DROP_EMBED, s0, s1);
            const unsigned bits = (u32_to_unit(p.x) < retain ? 1u : 0u) | (u32_to_unit(p.y) < retain ? 2u : 0u) |
                                  (u32_to_unit(p.z) < retain ? 4u : 0u) | (u32_to_unit(p.w) < retain ? 8u : 0u);
            dm |= bits << (4 * i);
        }
    }
    const bool pub = l == 0 && w.ct == 0 && ysl;
    GAS float* yin0 = m.yin0;
    auto afix = [&](int ci, float4 v, bool ok) -> float4 {
        if (dropping) {
            const unsigned nib = (dm >> (4 * ci)) & 15u;
            v.x *= (nib & 1u) ? inv_retain : 0.f; v.y *= (nib & 2u) ? inv_retain : 0.f;
            v.z *= (nib & 4u) ? inv_retain : 0.f; v.w *= (nib & 8u) ? inv_retain : 0.f;
        }
        if (pub && ok) st4(yin0 + (size_t)(m0 + sr) * IN + ks + 16 * ci + sc, v);
        return v;
    };
    GAS float* dst = m.vp + (size_t)s * B * D3;
    auto epi = [&](int row, int n, float v, float4) {
        if (row < M) dst[(size_t)row * D3 + n] = v;
    };
    gemm_tile2k_full<8>(m0, n0, Klen, aprov, bprov, m.zrow, epi, smem, afix);
}

// The gates behind k_gru_p1s (gru4rec.py:472-475): one quad of hidden units per thread -- K-slice partial sums of V added in slice order
// (candidate columns: the ny input slices; r / z columns: ny + nh), bias, r = sigmoid, Hr = H r, z = sigmoid; Vc / r / Hr / z to where
// k_gru_p2 and the backward kernels expect them.
__global__ __launch_bounds__(256) void k_gru_gate(const DevModel* __restrict__ mp, StepState* st, int l, int ny, int nh) {
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int D = m.D[l], D3 = 3 * D, nq = D >> 2, B = m.B, ns = ny + nh;
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int row = e / nq, d = 4 * (e - row * nq);
    if (row >= c.M) return;
    const GAS float* pp = m.vp + (size_t)row * D3 + d;
    const size_t ps = (size_t)B * D3;
    const GAS float* Bh = m.dense_p + m.offBh[l];
    float4 vc[8], vr[16], vz[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) vc[q] = ld4(pp + (size_t)min(q, ny - 1) * ps);
#pragma unroll
    for (int q = 0; q < 16; ++q) { vr[q] = ld4(pp + (size_t)min(q, ns - 1) * ps + D); vz[q] = ld4(pp + (size_t)min(q, ns - 1) * ps + 2 * D); }
    const float4 bc = ld4(Bh + d), br = ld4(Bh + D + d), bz = ld4(Bh + 2 * D + d);
    const size_t o = (size_t)row * D + d;
    const float4 h = ld4(m.H[l][c.g & 1] + o);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sr = sc, sz = sc;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q < ny) { sc.x += vc[q].x; sc.y += vc[q].y; sc.z += vc[q].z; sc.w += vc[q].w; }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < ns) {
            sr.x += vr[q].x; sr.y += vr[q].y; sr.z += vr[q].z; sr.w += vr[q].w;
            sz.x += vz[q].x; sz.y += vz[q].y; sz.z += vz[q].z; sz.w += vz[q].w;
        }
    const float4 r = make_float4(sigmoidf_(sr.x + br.x), sigmoidf_(sr.y + br.y), sigmoidf_(sr.z + br.z), sigmoidf_(sr.w + br.w));
    st4(m.Vc[l] + o, make_float4(sc.x + bc.x, sc.y + bc.y, sc.z + bc.z, sc.w + bc.w));
    st4(m.r[l] + o, r);
    st4(m.Hr[l] + o, make_float4(h.x * r.x, h.y * r.y, h.z * r.z, h.w * r.w));
    st4(m.z[l] + o, make_float4(sigmoidf_(sz.x + bz.x), sigmoidf_(sz.y + bz.y), sigmoidf_(sz.z + bz.z), sigmoidf_(sz.w + bz.w)));
}

// ---------------------------------------------------------------------------------------------------------------------------
// GRU backward, dy = dV Wx^T (K = 3D) as `nsb` K-slice partial sums dyp[slice][B][IN] (T.grad of gru4rec.py:472).  A = dV rows, B[n][k] =
// Wx[n][k]: both K-contiguous -> the LDS-DMA tile (gemm_tile3).  dy's consumer adds the slices up in slice order (DevModel::bbn): the
// lower layer's k_gru_bwd_pre, or -- layer 0 -- the row-finishing workgroups of k_dense_grad2 (embedding-dropout mask, per-occurrence
// Adagrad pieces dSx / dAx).
__global__ __launch_bounds__(256, 2) void k_gru_bwd_bw(const DevModel* __restrict__ mp, StepState* st, int l, int nsb, int kss) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D, B = m.B, M = c.M;
    const int nrt = (B + 63) >> 6;
    const WideItem w = wide_item(G4R_XCD_TILE(blockIdx.x, gridDim.x), nsb, nrt);
    const int m0 = w.rt * 64, n0 = w.ct * 64, ks = w.s * kss, Klen = min(kss, D3 - ks);
    if (m0 >= M) return;
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* dV = m.dV[l];
    GAS float* dst = m.dyp + (size_t)w.s * B * IN;
    auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? dV + (size_t)(m0 + r) * D3 + ks : nullptr; };
    auto brow = [&](int r) -> const GAS float* { return (n0 + r < IN) ? Wx + (size_t)(n0 + r) * D3 + ks : nullptr; };
    auto epi = [&](int row, int n, float v, float4) {
        if (row < M && n < IN) dst[(size_t)row * IN + n] = v;
    };
    gemm_tile3<3, 32, true>(m0, n0, Klen, arow, brow, m.zrow, NoPre(), epi, smem);
}

// The layer-0 input rows when k_gru_bwd_bw left dy as K-slice partial sums (DevModel::bbn[0] > 0): one quad of dy per thread (element e of
// the [B][IN / 4] quads) -- slices added in slice order, embedding-dropout mask, per-occurrence Adagrad pieces dSx / dAx (or the
// accumulator in place for an item that occurs once: see k_gru_bwd_b) -- ahead of the sparse row update that consumes them: as extra
// workgroups of k_dense_grad2 where that launch exists (no launch of its own: +2.8 us at configs[2] as one), else as k_finish_rows in
// front of the merged k_update.
__device__ __forceinline__ void finish_rows(const DevModel& m, const StepCtx& c, int e) {
    const int IN = m.IN[0], nq = IN >> 2, nsl = m.bbn[0], B = m.B;
    const int row = e / nq, c4 = 4 * (e - row * nq);
    if (row >= c.M) return;
    const int item = m.occ_idx[row];
    const GAS float* pp = m.dyp + (size_t)row * IN + c4;
    const size_t ps = (size_t)B * IN;
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = ld4(pp + (size_t)min(q, nsl - 1) * ps);
    GAS float* accT = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.accWy : m.accE;
    const float4 a0 = ld4_at(accT, (size_t)max(item, 0) * IN + c4, item >= 0);
    const int cnt1 = m.occ_fl[4 * (((m.embed_mode == G4R_EMBED_CONSTRAINED) ? (size_t)0 : (size_t)m.n_items) + max(item, 0)) + 2];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < nsl) { g.x += v[q].x; g.y += v[q].y; g.z += v[q].z; g.w += v[q].w; }
    if (m.drop_e > 0.f) {
        const float4 mk = drop_mult4(m.seed, (unsigned)c.g, G4R_STREAM_DROP_EMBED, row, c4 >> 2, 1.0f - m.drop_e);
        g.x *= mk.x; g.y *= mk.y; g.z *= mk.z; g.w *= mk.w;
    }
    const size_t o = (size_t)row * IN + c4;
    const float lr = m.lr;
    const bool generic = m.generic != 0;
    const float4 an = make_float4(a0.x + G4R_MUT_ACC(g.x * g.x), a0.y + G4R_MUT_ACC(g.y * g.y), a0.z + G4R_MUT_ACC(g.z * g.z), a0.w + G4R_MUT_ACC(g.w * g.w));
    const float4 stp = generic ? g : make_float4(G4R_MUT_STEP(lr * g.x * frsq(an.x + G4R_EPS_ADAGRAD)), G4R_MUT_STEP(lr * g.y * frsq(an.y + G4R_EPS_ADAGRAD)),
                                                 G4R_MUT_STEP(lr * g.z * frsq(an.z + G4R_EPS_ADAGRAD)), G4R_MUT_STEP(lr * g.w * frsq(an.w + G4R_EPS_ADAGRAD)));
    st4(G4R_DSX(m, c.g) + o, stp);
    if (!generic && cnt1 == 1 && item >= 0) st4(accT + (size_t)item * IN + c4, an);
    else st4(m.dAx + o, an);
}
__global__ __launch_bounds__(256) void k_finish_rows(const DevModel* __restrict__ mp, StepState* st) {
    finish_rows(*mp, load_ctx(st), (int)blockIdx.x * 256 + (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Dense GRU gradients on 64 x 64 tiles (contraction over the batch: both operands K-major -> gemm_tile2k):
//   dWx = yin^T dV ; dWh = (H r)^T dV[:, :D] ; dWrz = H^T dV[:, D:] ; dBh = colsum(dV)   (descriptor entries with nrows == 1: plain
// column sums of 64 columns) with the dense Adagrad(+momentum) update (gru4rec.py:330-334,390-406) fused into the epilogue on a
// single GPU, or the gradient -> dense_g for the all-reduce.  Against the 32 x 32 tiles of k_update (dense_grad_tile): half the
// operand bytes per output, the 32x32x2 MFMA, operands two chunks ahead.  Runs as a launch of its own in front of the sparse row
// update (at wide layers the two roles of k_update did not overlap anyway: DESIGN.md section 6).
__global__ __launch_bounds__(256, 2) void k_dense_grad2(const DevModel* __restrict__ mp, StepState* st, const DenseTile* __restrict__ tiles_, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const GAS DenseTile* tiles = (const GAS DenseTile*)tiles_;
    const StepCtx c = load_ctx(st);
    if ((int)blockIdx.x >= ntiles) { finish_rows(m, c, ((int)blockIdx.x - ntiles) * 256 + (int)threadIdx.x); return; }      // row-finishing workgroups behind the tiles
    const DenseTile tl = tiles[G4R_XCD_TILE(blockIdx.x, ntiles)];
    const int M = c.M, tid = threadIdx.x;
    const float lr = m.lr, momc = m.mom, lmbd = m.lmbd;
    const int inplace = m.apply_dense_inplace;
    GAS float *dp = m.dense_p, *dacc = m.dense_acc, *dvel = m.dense_vel, *dg = m.dense_g;
    const GAS float* dV = tl.dV;
    if (tl.nrows == 1) {
        // bias row: column sums of dV over the batch for 64 columns; thread (column tid & 63, row group tid >> 6)
        const int cl = tid & 63, grp = tid >> 6, col = tl.c0 + cl;
        const bool cok = col < tl.ncols;
        const size_t off = (size_t)tl.base + (cok ? col : 0);
        const bool ok0 = cok && inplace != 0 && tid < 64;
        const float a0 = ldf_at(dacc, off, ok0), p0 = ldf_at(dp, off, ok0), v0 = ldf_at(dvel, off, ok0 && momc > 0.f);
        float s = 0.f;
        for (int b0 = grp; b0 < M; b0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ldf_if(dV, (size_t)min(b0 + 4 * u, M - 1) * tl.ldv + tl.coff + (cok ? col : 0), cok && b0 + 4 * u < M);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        smem[grp * 64 + cl] = s;
        __syncthreads();
        if (tid < 64 && cok) {
            const float g = (smem[cl] + smem[64 + cl]) + (smem[128 + cl] + smem[192 + cl]);
            if (!inplace) { dg[off] = g; return; }
            const float acc = a0 + G4R_MUT_DACC(g * g);            // gru4rec.py:330-334,390-406
            dacc[off] = acc;
            const float gs = g * frsq(acc + G4R_EPS_ADAGRAD);
            if (momc > 0.f) {
                const float v = momc * v0 - lr * (gs + lmbd * p0);
                dvel[off] = v;
                dp[off] = p0 + v;
            } else {
                dp[off] = p0 * (1.0f - lr * lmbd) - lr * gs;
            }
        }
        return;
    }
    const GAS float* X = tl.gather ? (const GAS float*)m.yin0 : ((c.g & 1) ? tl.X1 : tl.X0);
    auto aptr = [&](int kk, int kr, int cc) -> const GAS float* {       // X[b = kk + kr][r0 + cc ..]
        return (kk + kr < M && tl.r0 + cc < tl.nrows) ? X + (size_t)(kk + kr) * tl.ldx + tl.r0 + cc : nullptr;
    };
    auto bptr = [&](int kk, int kr, int cc) -> const GAS float* {       // dV[b = kk + kr][coff + c0 + cc ..]
        return (kk + kr < M && tl.c0 + cc < tl.ncols) ? dV + (size_t)(kk + kr) * tl.ldv + tl.coff + tl.c0 + cc : nullptr;
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
    gemm_tile2k<true, false>(tl.r0, tl.c0, M, aptr, bptr, m.zrow, pre, epi, smem);
}
