// g4r_bwd_kernels.cuh -- part of g4r_step_kernels.cuh (included there, in order; needs its prelude).  Holds backward of the step: k_score_bwd / k_score_bwd2, k_gru_bwd_pre / _a / _b, k_gru_bwd_fused.
#pragma once
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
            float step = (p.y != 0.f) ? G4R_MUT_ROW(n, G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD))) : 0.f;
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
            float step = (p.y != 0.f) ? G4R_MUT_ROW(n, G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD))) : 0.f;
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

// The Adagrad rule for the item rows of the scoring backward, for launches whose role A left RAW gradient rows in the step plane
// (k_score_bmt): g = dSy[n][:] -> acc' = acc + g^2, step = lr g / sqrt(acc' + eps); the step replaces g, acc' goes to the accumulator table in
// place when the row's item occurs once in the step and to dAy[n] otherwise (k_score_bwd2's role-A epilogue; gru4rec.py:335-340).  A wave per
// row, float4 per lane (D <= 512); workgroup `wg` of `nwg` takes the 16-row chunks wg, wg + nwg, ... (4 waves x 4 rows, every load of a chunk
// in flight at once: the chain item -> accumulator row -> stores is latency, so the host gives every chunk a workgroup of its own).
__device__ __forceinline__ void score_fin_rows(const DevModel& m, long long g_, int wg, int nwg) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int nw = 4;                   // (the launches that carry this role have 4 or 8 waves: the row partition must not depend on that)
    if (w >= nw) return;
    const int N = m.N, D = m.Dtop, D4 = D >> 2;
    GAS float* dSy = G4R_DSY(m, g_);
    GAS float *dAy = m.dAy, *accWy = m.accWy;
    const GAS int *col_item = m.col_item, *occ_fl = m.occ_fl;
    const float lr = m.lr;
    const bool generic = m.generic != 0;
    constexpr int UN = 4;
    for (int r0 = (wg * nw + w) * UN; r0 < N; r0 += nwg * nw * UN) {
        int item[UN], cnt[UN];
        float4 g[UN][2], a0[UN][2];
#pragma unroll
        for (int u = 0; u < UN; ++u) item[u] = col_item[min(r0 + u, N - 1)];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = min(r0 + u, N - 1);
            cnt[u] = occ_fl[4 * (size_t)max(item[u], 0) + 2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int d4 = min(lane + 64 * q, D4 - 1);
                g[u][q] = ld4(dSy + (size_t)n * D + 4 * d4);
                a0[u][q] = ld4(accWy + (size_t)max(item[u], 0) * D + 4 * d4);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = r0 + u;
            if (n >= N) continue;
            const bool ok = item[u] >= 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int d4 = lane + 64 * q;
                if (d4 >= D4) continue;
                const float gv[4] = {g[u][q].x, g[u][q].y, g[u][q].z, g[u][q].w}, av[4] = {a0[u][q].x, a0[u][q].y, a0[u][q].z, a0[u][q].w};
                float sv[4], nv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    nv[e] = av[e] + G4R_MUT_ACC(gv[e] * gv[e]);
                    sv[e] = ok ? G4R_MUT_ROW(n, G4R_MUT_STEP(lr * gv[e] * frsq(nv[e] + G4R_EPS_ADAGRAD))) : 0.f;
                    if (generic) sv[e] = ok ? gv[e] : 0.f;
                }
                st4(dSy + (size_t)n * D + 4 * d4, make_float4(sv[0], sv[1], sv[2], sv[3]));
                if (!generic && ok && cnt[u] == 1) st4(accWy + (size_t)item[u] * D + 4 * d4, make_float4(nv[0], nv[1], nv[2], nv[3]));
                else st4(dAy + (size_t)n * D + 4 * d4, make_float4(nv[0], nv[1], nv[2], nv[3]));
            }
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
// nfin > 0 (top layer behind k_score_bmt): the grid has nfin extra rows of workgroups that finish the item rows of the scoring backward
// (score_fin_rows: 16 rows each, one batch of loads) next to this launch's 128-odd tiles
template <int NTH, int BK>
__global__ __launch_bounds__(NTH) void k_gru_bwd_a(const DevModel* __restrict__ mp, StepState* st, int l, int nfin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const StepCtx c = load_ctx(st);
    if (nfin > 0 && (int)blockIdx.y >= (int)gridDim.y - nfin) {
        score_fin_rows(m, c.g, ((int)blockIdx.y - ((int)gridDim.y - nfin)) * (int)gridDim.x + (int)blockIdx.x, nfin * (int)gridDim.x);
        return;
    }
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
