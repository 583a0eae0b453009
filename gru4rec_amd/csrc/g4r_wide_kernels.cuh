// GRU kernels for WIDE layers (D a multiple of 64, >= 256 units; gfx950): the five GEMMs of a layer's step and its dense gradients
// on 64 x 64 tiles of v_mfma_f32_32x32x2_f32 (g4r_gemm.cuh: gemm_tile2k / gemm_tile3) with the K range of a tile split over several
// workgroups and joined inside the launch by the tile's last arriver (SplitKJoin).
//
// Why (BASELINE configs[2]: B = 240, D = 512; profiles/r04_*): the round-1 kernels (k_gru_p1<64,256>, k_gru_p2, k_gru_bwd_a / _b,
// 32 x 32 dense-gradient tiles) keep the whole K range in one workgroup, which makes few and fat workgroups -- k_gru_p1 192 of them,
// 384 KB of operands each, on 256 CUs -- and one CU only pulls ~36 GB/s out of L2 whatever it does ("the fetch
// path"): 25.6 us for 629 MFLOP, 15.6 % of the fp32 MFMA peak.  Here a workgroup owns 64 x 64 outputs over a K SLICE of 128-256:
// 64-128 KB of operands, >= 256 workgroups for the big products, half the operand bytes in total (1 / 64 + 1 / 64 per flop instead
// of 1 / 32 + 1 / 64).  The reference's math is unchanged (gru4rec.py:471-479 and its T.grad, :383-384); only the fp32 summation
// order differs (k-ordered inside a slice, slices added in slice order: deterministic).
#pragma once
#include "g4r_step_kernels.cuh"

#ifndef G4R_WIDE_NST
#define G4R_WIDE_NST 3      // ring depth (stages of 32 k, 16 KB each) of the LDS-DMA fed wide kernels
#endif

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
// GRU phase 1 (training), gru4rec.py:472-475:  V[B, 3D] = [y | H] [Wx ; 0 | Wrz] + Bh ; Vc = V[:, :D], r = sigmoid(V[:, D:2D]),
// Hr = H r, z = sigmoid(V[:, 2D:]).  K slices: `ny` slices of `kys` input units (A = the layer's input rows: gathered table rows
// with embedding dropout for layer 0), then `nh` slices of `khs` hidden units (A = H; only for the r / z columns: the candidate
// columns have no hidden part here -- theirs is (H r) Wh, phase 2).  B = rows of Wx / Wrz ([k][n], K-major).
// Embedding dropout: the Philox masks of a thread's staging slots (one quad per 16-deep chunk: row tid >> 2, k offset 4 (tid & 3))
// are drawn up front, next to the first operand requests, as 4 bits per chunk.
// SLAB: no join -- every slice stores its partial tile to vp[slice][B][3D] and k_gru_gate (the next launch) adds the slices up and
// applies the gates; the kernel then is a bare GEMM: no epilogue operands, no join buffers, half the registers (more workgroups per CU).
template <bool SLAB>
__device__ __forceinline__ void gru_p1w_body(const DevModel* __restrict__ mp, StepState* st, int l, int first, float* ws, unsigned* cnt,
                                             int ny, int nh, int kys, int khs, float* smem) {
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
    const int nsplit = cand ? ny : ny + nh;
    WideItem w = wide_item(cand ? idx : idx - base_c, nsplit, nrt);
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
    if (m0 >= M) return;      // (all slices of the tile: its counter stays untouched)
    const bool ysl = s < ny;
    const int ks = ysl ? s * kys : (s - ny) * khs;
    const int Klen = ysl ? min(kys, IN - ks) : min(khs, D - ks);
    const GAS float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const GAS float* ysrc = l > 0 ? m.hd[l - 1] : nullptr;
    const GAS float* Hcur = m.H[l][g & 1];
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wrz = m.dense_p + m.offWrz[l];
    const GAS float* Bh = m.dense_p + m.offBh[l];
    auto aprov = [&](int r) -> const GAS float* {
        const int row = m0 + r;
        if (row >= M) return nullptr;
        if (!ysl) return Hcur + (size_t)row * D + ks;
#if defined(G4R_P1S_DBG) && (G4R_P1S_DBG & 2)
        if (l == 0) return Hcur + (size_t)row * D + ks;
#endif
        if (l == 0) return table + (size_t)gidx[row] * IN + ks;
        return ysrc + (size_t)row * IN + ks;
    };
    auto bprov = [&](int kk, int kr, int kc) -> const GAS float* {
        const int k = kk + kr;
        if (k >= Klen) return nullptr;
        return ysl ? Wx + (size_t)(ks + k) * D3 + n0 + kc : Wrz + (size_t)(ks + k) * (2 * D) + (n0 - D) + kc;
    };
    // dropout bits of this thread's staging slots: chunk i < 16 in dm0, else in dm1 (host: a y slice is <= 512 units)
    const int sr = tid >> 2, sc = 4 * (tid & 3);
#if defined(G4R_P1S_DBG) && (G4R_P1S_DBG & 4)
    const bool dropping = false;
#else
    const bool dropping = ysl && l == 0 && m.drop_e > 0.f;
#endif
    const float retain = 1.0f - m.drop_e, inv_retain = 1.0f / retain;
    unsigned long long dm0 = ~0ull, dm1 = ~0ull;
    if (dropping) {
        dm0 = 0ull; dm1 = 0ull;
        const unsigned s0 = (unsigned)m.seed, s1 = (unsigned)(m.seed >> 32);
        const int nch = Klen >> 4;
        for (int i = 0; i < nch; ++i) {
            const Philox4 p = philox4x32_10((unsigned)((ks + 16 * i + sc) >> 2), (unsigned)(m0 + sr), g, G4R_STREAM_DROP_EMBED, s0, s1);
            const unsigned long long bits = (u32_to_unit(p.x) < retain ? 1ull : 0ull) | (u32_to_unit(p.y) < retain ? 2ull : 0ull) |
                                            (u32_to_unit(p.z) < retain ? 4ull : 0ull) | (u32_to_unit(p.w) < retain ? 8ull : 0ull);
            if (i < 16) dm0 |= bits << (4 * i); else dm1 |= bits << (4 * (i - 16));
        }
    }
    // column tile 0 publishes the (masked) layer-0 input rows: the dense-gradient tiles read them back (dWx = yin^T dV)
    const bool pub = l == 0 && w.ct == 0 && ysl;
    GAS float* yin0 = m.yin0;
    auto afix = [&](int ci, float4 v, bool ok) -> float4 {
        if (dropping) {
            const unsigned nib = (unsigned)((ci < 16 ? dm0 >> (4 * ci) : dm1 >> (4 * (ci - 16))) & 15ull);
            v.x *= (nib & 1u) ? inv_retain : 0.f; v.y *= (nib & 2u) ? inv_retain : 0.f;
            v.z *= (nib & 4u) ? inv_retain : 0.f; v.w *= (nib & 8u) ? inv_retain : 0.f;
        }
        if (pub && ok) st4(yin0 + (size_t)(m0 + sr) * IN + ks + 16 * ci + sc, v);
        return v;
    };
    GAS float *Vc = m.Vc[l], *zb = m.z[l], *Hrb = m.Hr[l], *rb = m.r[l];
    auto pre = [&](int row, int n) -> float4 {      // bias and (r block) the hidden value
        const bool ok = row < M;
        return make_float4(ldf_at(Bh, n, ok), ldf_at(Hcur, (size_t)row * D + (n - D), ok && n >= D && n < 2 * D), 0.f, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M) return;
        v += p.x;
        if (n < D) { Vc[(size_t)row * D + n] = v; return; }
        if (n < 2 * D) {
            const size_t o = (size_t)row * D + (n - D);
            const float rr = sigmoidf_(v);
            rb[o] = rr;
            Hrb[o] = p.y * rr;
            return;
        }
        zb[(size_t)row * D + (n - 2 * D)] = sigmoidf_(v);
    };
    if constexpr (SLAB) {
        GAS float* dst = m.vp + (size_t)s * B * D3;
        auto epis = [&](int row, int n, float v, float4) {
#if defined(G4R_P1S_DBG) && (G4R_P1S_DBG & 1)
            if (v == 123.456f)
#endif
            if (row < M) dst[(size_t)row * D3 + n] = v;
        };
        if (Klen <= 128) gemm_tile2k_full<8>(m0, n0, Klen, aprov, bprov, m.zrow, epis, smem, afix);      // the default geometry: the whole slice in flight at once
        else gemm_tile2k<false, true>(m0, n0, Klen, aprov, bprov, m.zrow, NoPre(), epis, smem, nullptr, NoJoin(), afix);
    } else {
        const SplitKJoin join = {ws, cnt, w.ct * nrt + w.rt, s, nsplit, ny + nh};
        gemm_tile2k<false, false>(m0, n0, Klen, aprov, bprov, m.zrow, pre, epi, smem, nullptr, join, afix);
    }
}
__global__ __launch_bounds__(256, 2) void k_gru_p1w(const DevModel* __restrict__ mp, StepState* st, int l, int first, float* ws, unsigned* cnt,
                                                    int ny, int nh, int kys, int khs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gru_p1w_body<false>(mp, st, l, first, ws, cnt, ny, nh, kys, khs, smem);
}
__global__ __launch_bounds__(256, 4) void k_gru_p1s(const DevModel* __restrict__ mp, StepState* st, int l, int first, int ny, int nh, int kys, int khs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gru_p1w_body<true>(mp, st, l, first, nullptr, nullptr, ny, nh, kys, khs, smem);
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
// GRU phase 2 (training), gru4rec.py:474-479:  c = act(Hr Wh + Vc), h = (1 - z) H + z c, hidden dropout, reset switch -> next H.
// A = Hr rows (K-contiguous), B = rows of Wh ([k][n]); K = D in `nsp` slices of `kss`.
__global__ __launch_bounds__(256, 2) void k_gru_p2w(const DevModel* __restrict__ mp, StepState* st, int l, float* ws, unsigned* cnt, int nsp, int kss) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int D = m.D[l], B = m.B, M = c.M;
    const unsigned g = (unsigned)c.g;
    const int nrt = (B + 63) >> 6;
    const WideItem w = wide_item(G4R_XCD_TILE(blockIdx.x, gridDim.x), nsp, nrt);
    const int m0 = w.rt * 64, n0 = w.ct * 64, ks = w.s * kss, Klen = min(kss, D - ks);
    if (m0 >= M) return;
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float *Hcur = m.H[l][g & 1], *Vc = m.Vc[l], *zb = m.z[l], *Hrb = m.Hr[l];
    GAS float *Hnext = m.H[l][(g + 1) & 1], *hout = m.hd[l], *cl = m.c[l];
    const GAS unsigned char* rst = m.reset + c.t * B;
    const float retain_h = 1.0f - m.drop_h, drop_h = m.drop_h, hp0 = m.ha_p0, hp1 = m.ha_p1;
    const int hact = m.hidden_act;
    const unsigned long long seed = m.seed;
    auto aprov = [&](int r) -> const GAS float* { return (m0 + r < M) ? Hrb + (size_t)(m0 + r) * D + ks : nullptr; };
    auto bprov = [&](int kk, int kr, int kc) -> const GAS float* {
        return (kk + kr < Klen) ? Wh + (size_t)(ks + kk + kr) * D + n0 + kc : nullptr;
    };
    auto pre = [&](int row, int n) -> float4 {
        const bool ok = row < M;
        const size_t o = (size_t)row * D + n;
        return make_float4(ldf_at(Vc, o, ok), ldf_at(zb, o, ok), ldf_at(Hcur, o, ok), rst[ok ? row : 0] ? 1.f : 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row >= M) return;
        const size_t o = (size_t)row * D + n;
        const float cc = act_fwd(hact, hp0, hp1, v + p.x);
        float h = (1.0f - p.y) * p.z + p.y * cc;
        if (drop_h > 0.f) h *= drop_mult(seed, g, G4R_STREAM_DROP_HIDDEN + l, row, n, retain_h);
        cl[o] = cc;
        hout[o] = h;
        Hnext[o] = p.w != 0.f ? 0.f : h;
    };
    const SplitKJoin join = {ws, cnt, w.ct * nrt + w.rt, w.s, nsp, nsp};
    gemm_tile2k<false, false>(m0, n0, Klen, aprov, bprov, m.zrow, pre, epi, smem, nullptr, join);
}

// ---------------------------------------------------------------------------------------------------------------------------
// GRU backward, dr' = (da Wh^T) H r (1 - r) -> dV[:, D:2D]   (T.grad of gru4rec.py:474).  A = da = dV[:, :D] rows, B[n][k] = Wh[n][k]:
// both K-contiguous -> the LDS-DMA tile (gemm_tile3); K = D in `nsp` slices of `kss` (multiples of 32).
__global__ __launch_bounds__(256, 2) void k_gru_bwd_aw(const DevModel* __restrict__ mp, StepState* st, int l, float* ws, unsigned* cnt, int nsp, int kss) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int D = m.D[l], D3 = 3 * D, B = m.B, M = c.M;
    const int nrt = (B + 63) >> 6;
    const WideItem w = wide_item(G4R_XCD_TILE(blockIdx.x, gridDim.x), nsp, nrt);
    const int m0 = w.rt * 64, n0 = w.ct * 64, ks = w.s * kss, Klen = min(kss, D - ks);
    if (m0 >= M) return;
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float* Hcur = m.H[l][c.g & 1];
    const GAS float* rl = m.r[l];
    GAS float* dV = m.dV[l];
    auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? dV + (size_t)(m0 + r) * D3 + ks : nullptr; };
    auto brow = [&](int r) -> const GAS float* { return (n0 + r < D) ? Wh + (size_t)(n0 + r) * D + ks : nullptr; };
    auto pre = [&](int row, int n) -> float4 {
        const bool ok = row < M;
        const size_t o = (size_t)row * D + n;
        return make_float4(ldf_at(rl, o, ok), ldf_at(Hcur, o, ok), 0.f, 0.f);
    };
    auto epi = [&](int row, int n, float v, float4 p) {
        if (row < M) dV[(size_t)row * D3 + D + n] = v * p.y * p.x * (1.f - p.x);
    };
    const SplitKJoin join = {ws, cnt, w.ct * nrt + w.rt, w.s, nsp, nsp};
    gemm_tile3<G4R_WIDE_NST, 32, false>(m0, n0, Klen, arow, brow, m.zrow, pre, epi, smem, nullptr, join);
}

// ---------------------------------------------------------------------------------------------------------------------------
// GRU backward, dy = dV Wx^T (K = 3D) -> layer 0: through the embedding-dropout mask into the per-occurrence Adagrad pieces of the
// input rows (dSx, dAx / accumulator in place: see k_gru_bwd_b); upper layers: the lower layer's dh.  A = dV rows, B[n][k] =
// Wx[n][k]: both K-contiguous -> gemm_tile3, `nsb` slices of `kss`.  The last arriver turns its accumulators around through LDS so
// that a thread finishes four QUADS of consecutive columns (16-byte loads / stores of the accumulator rows, one Philox draw per quad
// instead of one per element).
// slabs != 0: no join at all -- every slice stores its partial tile to dyp[slice][B][IN] and the kernel that consumes dy anyway adds the
// slices up behind the kernel boundary (DevModel::bbn): the in-launch join measured ~7 us of write-through publish, ticket and re-read,
// more than the K loop of a slice (profiles/r05_experiments.md)
__global__ __launch_bounds__(256, 2) void k_gru_bwd_bw(const DevModel* __restrict__ mp, StepState* st, int l, float* ws, unsigned* cnt, int nsb, int kss, int slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D, B = m.B, M = c.M;
    const int nrt = (B + 63) >> 6;
    const WideItem w = wide_item(G4R_XCD_TILE(blockIdx.x, gridDim.x), nsb, nrt);
    const int m0 = w.rt * 64, n0 = w.ct * 64, ks = w.s * kss, Klen = min(kss, D3 - ks);
    if (m0 >= M) return;
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* dV = m.dV[l];
    if (slabs) {
        GAS float* dst = m.dyp + (size_t)w.s * B * IN;
        auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? dV + (size_t)(m0 + r) * D3 + ks : nullptr; };
        auto brow = [&](int r) -> const GAS float* { return (n0 + r < IN) ? Wx + (size_t)(n0 + r) * D3 + ks : nullptr; };
        auto epi = [&](int row, int n, float v, float4) {
            if (row < M && n < IN) dst[(size_t)row * IN + n] = v;
        };
        gemm_tile3<G4R_WIDE_NST, 32, true>(m0, n0, Klen, arow, brow, m.zrow, NoPre(), epi, smem);
        return;
    }
    // epilogue ownership: rows m0 + (tid >> 4) + 16 i (i = 0..3), columns n0 + 4 (tid & 15) .. + 3
    const int er = tid >> 4, ec = n0 + 4 * (tid & 15);
    const bool cok = ec < IN;
    GAS float* accT = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.accWy : m.accE;
    const GAS int* occ_fl = m.occ_fl + 4 * ((m.embed_mode == G4R_EMBED_CONSTRAINED) ? (size_t)0 : (size_t)m.n_items);
    int item[4], pcnt[4];
    float4 pacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + er + 16 * i;
        item[i] = (l == 0 && row < M) ? m.occ_idx[row] : -1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // pre-step accumulator quad of the input item's row and its occurrence count (layer 0)
        pacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pcnt[i] = 0;
        if (l == 0) {
            pacc[i] = ld4_at(accT, (size_t)max(item[i], 0) * IN + ec, item[i] >= 0 && cok);
            pcnt[i] = occ_fl[4 * (size_t)max(item[i], 0) + 2];
        }
    }
    auto arow = [&](int r) -> const GAS float* { return (m0 + r < M) ? dV + (size_t)(m0 + r) * D3 + ks : nullptr; };
    auto brow = [&](int r) -> const GAS float* { return (n0 + r < IN) ? Wx + (size_t)(n0 + r) * D3 + ks : nullptr; };
    const float lr = m.lr, drop_e = m.drop_e;
    const bool generic = m.generic != 0;
    const unsigned long long seed = m.seed;
    GAS float *dSx = m.dSx, *dAx = m.dAx, *dylo = (l > 0) ? m.dyl[l - 1] : nullptr;
    const SplitKJoin sj = {ws, cnt, w.ct * nrt + w.rt, w.s, nsb, nsb};
    auto join = [&](f32x16& acc, float* sm) -> bool {
        if (!sj(acc, sm)) return false;
        constexpr int LDT = 68;      // [64][68] floats: 16-byte aligned rows
        const int wm = wid >> 1, wn = wid & 1, l32 = lane & 31, lh = lane >> 5;
        __syncthreads();             // everybody is past the operand ring / the ticket word
#pragma unroll
        for (int j = 0; j < 16; ++j) sm[(wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3)) * LDT + wn * 32 + l32] = acc[j];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl_ = er + 16 * i, row = m0 + rl_;
            if (row >= M || !cok) continue;
            float4 v = *reinterpret_cast<const float4*>(sm + rl_ * LDT + 4 * (tid & 15));
            const size_t o = (size_t)row * IN + ec;
            if (l > 0) { st4(dylo + o, v); continue; }
            if (drop_e > 0.f) {
                const float4 mk = drop_mult4(seed, (unsigned)c.g, G4R_STREAM_DROP_EMBED, row, ec >> 2, 1.0f - drop_e);
                v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
            }
            const float4 an = make_float4(pacc[i].x + G4R_MUT_ACC(v.x * v.x), pacc[i].y + G4R_MUT_ACC(v.y * v.y),
                                          pacc[i].z + G4R_MUT_ACC(v.z * v.z), pacc[i].w + G4R_MUT_ACC(v.w * v.w));
            const float4 stp = generic ? v : make_float4(G4R_MUT_STEP(lr * v.x * frsq(an.x + G4R_EPS_ADAGRAD)), G4R_MUT_STEP(lr * v.y * frsq(an.y + G4R_EPS_ADAGRAD)),
                                                         G4R_MUT_STEP(lr * v.z * frsq(an.z + G4R_EPS_ADAGRAD)), G4R_MUT_STEP(lr * v.w * frsq(an.w + G4R_EPS_ADAGRAD)));
            st4(dSx + o, stp);
            // single-occurrence item: new accumulator in place (see k_score_bwd), else through dA and the update kernel's owner wave
            if (!generic && pcnt[i] == 1 && item[i] >= 0) st4(accT + (size_t)item[i] * IN + ec, an);
            else st4(dAx + o, an);
        }
        return false;                // (the tile's generic per-element epilogue is not used)
    };
    auto epi = [&](int, int, float, float4) {};
    gemm_tile3<G4R_WIDE_NST, 32, true>(m0, n0, Klen, arow, brow, m.zrow, NoPre(), epi, smem, nullptr, join);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Dense GRU gradients on 64 x 64 tiles (contraction over the batch: both operands K-major -> gemm_tile2k):
//   dWx = yin^T dV ; dWh = (H r)^T dV[:, :D] ; dWrz = H^T dV[:, D:] ; dBh = colsum(dV)   (descriptor entries with nrows == 1: plain
// column sums of 64 columns) with the dense Adagrad(+momentum) update (gru4rec.py:330-334,390-406) fused into the epilogue on a
// single GPU, or the gradient -> dense_g for the all-reduce.  Against the 32 x 32 tiles of k_update (dense_grad_tile): half the
// operand bytes per output, the 32x32x2 MFMA, operands two chunks ahead.  Runs as a launch of its own in front of the sparse row
// update (at wide layers the two roles of k_update did not overlap anyway: DESIGN.md section 6).
// Workgroups [ntiles, gridDim) finish the layer-0 input rows when k_gru_bwd_bw left dy as K-slice partial sums (DevModel::bbn[0] > 0):
// one quad of dy per thread -- slices added in slice order, embedding-dropout mask, per-occurrence Adagrad pieces dSx / dAx (or the
// accumulator in place for an item that occurs once: see k_gru_bwd_b) -- ahead of the sparse row update that consumes them.
__global__ __launch_bounds__(256, 2) void k_dense_grad2(const DevModel* __restrict__ mp, StepState* st, const DenseTile* __restrict__ tiles_, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const GAS DenseTile* tiles = (const GAS DenseTile*)tiles_;
    const StepCtx c = load_ctx(st);
    if ((int)blockIdx.x >= ntiles) {
        const int IN = m.IN[0], nq = IN >> 2, nsl = m.bbn[0], B = m.B;
        const int e = ((int)blockIdx.x - ntiles) * 256 + (int)threadIdx.x;
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
        st4(m.dSx + o, stp);
        if (!generic && cnt1 == 1 && item >= 0) st4(accT + (size_t)item * IN + c4, an);
        else st4(m.dAx + o, an);
        return;
    }
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
