// Training-step kernels (gfx950).  One mini-batch step of GRU4Rec's session-parallel loop
// (reference: the Theano function built at gru4rec.py:572-584 and called at :623) is 6+n_layers*2
// launches:
//   k_gru_fwd (per layer)  gather + dropout + GRU step                  gru4rec.py:438-479
//   k_score_fwd            gathered-row scoring GEMM (fp32 MFMA)         gru4rec.py:480-495
//   k_loss_rows            final activation + loss + d/ds per row        gru4rec.py:193-248,496
//   k_score_bwd            dSy = ds^T h , dSBy , split-K partials of ds Sy   (T.grad, :383-384)
//   k_gru_bwd_rows (layer) GRU backward, row-local part
//   k_dense_grad           batch contractions dWx/dWh/dWrz/dBh (+ fused dense Adagrad) :390-406
//   k_sparse_update        per-occurrence Adagrad on the touched Wy/By/E rows :407-431 + step bookkeeping
#pragma once
#include "g4r_device.cuh"

#define GRU_NW 8          // waves per workgroup in the GRU row kernels
#define GRU_ROWS 16       // batch rows per workgroup (one MFMA tile high)
// The GRU row kernels are compiled for three width classes (max(D, IN) <= 128 / 256 / 512):
//   T1 = 16-col tiles per wave over 3D columns, T2 = tiles per wave over D (or IN) columns,
//   U1/U2 = k-steps whose operand loads are batched ahead of the MFMAs (see tile_gemm_rows).

struct StepCtx { long long t, g; int M; };

__device__ __forceinline__ StepCtx load_ctx_first(const DevModel& m) {
    StepCtx c;
    c.t = m.st->t_a;
    c.g = m.st->g_a;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { m.st->t_b = c.t; m.st->g_b = c.g; }
    c.M = m.Mplan[c.t];
    return c;
}
__device__ __forceinline__ StepCtx load_ctx(const DevModel& m) {
    StepCtx c;
    c.t = m.st->t_b;
    c.g = m.st->g_b;
    c.M = m.Mplan[c.t];
    return c;
}

// ---------------------------------------------------------------------------------------------
// GRU forward for one layer, 16 batch rows per workgroup, all columns.  Used for training
// (train = 1: step context from device state, dropout, reset switch, activations saved) and for
// prediction (train = 0: explicit arguments, gru4rec.py:433 predict=True).
struct GruFwdPredict {
    GP(const int) in_idx;   // device, layer 0 gather indices
    GP(const float) ysrc;   // layer > 0 input rows
    GP(const float) Hcur;
    GP(float) Hnext;
    GP(float) hout;         // [rows][D]
    int M;
};

template <int T1, int U1, int T2, int U2>
__global__ __launch_bounds__(GRU_NW * 64) void k_gru_fwd(const DevModel* __restrict__ mp, int l, int train, int first, GruFwdPredict pa) {
    const DevModel& m = *mp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int D = m.D[l], IN = m.IN[l], D3 = 3 * D;
    const int ldy = IN + 2, ldh = D + 2, ldv = D3 + 2;
    float* sH = smem;                               // [16][ldh]  H, later H*r
    float* sY = smem + GRU_ROWS * ldh;              // [16][ldy]  layer input (phase 1)
    float* sV = sY;                                 // [16][ldv]  V(+G) (after phase 1; aliases sY)
    long long t = 0, g = 0;
    int M;
    const GAS float *Hcur, *ysrc = nullptr;
    GAS float* Hnext;
    const GAS int* gidx = nullptr;
    if (train) {
        StepCtx c = first ? load_ctx_first(m) : load_ctx(m);
        t = c.t; g = c.g; M = c.M;
        Hcur = m.H[l][g & 1];
        Hnext = m.H[l][(g + 1) & 1];
        if (l == 0) gidx = m.in_idx + t * m.B; else ysrc = m.hd[l - 1];
    } else {
        M = pa.M; Hcur = pa.Hcur; Hnext = pa.Hnext; gidx = pa.in_idx; ysrc = pa.ysrc;   // kernel arguments: already global
    }
    G4R_TICK(m, 0, 0);
    if (m.dbgclk && train && first && blockIdx.x == 0 && threadIdx.x < 4) m.dbgclk[32 + threadIdx.x] = 0;   // per-step debug stats
    const int r0 = blockIdx.x * GRU_ROWS;
    if (train && l == 0 && tid < GRU_ROWS) {
        const int row = r0 + tid;
        if (row < m.B) m.occ_idx[row] = row < M ? gidx[row] : -1;
    }
    if (r0 >= M) return;
    const GAS float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const float retain_e = 1.0f - m.drop_e, retain_h = 1.0f - m.drop_h;
    // ---- stage input rows (gather + embedding dropout) and hidden rows
    for (int e = tid; e < GRU_ROWS * (IN >> 2); e += GRU_NW * 64) {
        const int i = e / (IN >> 2), c4 = e - i * (IN >> 2), row = r0 + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) {
            const GAS float* src = (l == 0) ? table + (size_t)gidx[row] * IN : ysrc + (size_t)row * IN;
            v = ld4(src + 4 * c4);
            if (train && l == 0) {
                if (m.drop_e > 0.f) {
                    const float4 mk = drop_mult4(m.seed, (unsigned)g, G4R_STREAM_DROP_EMBED, row, c4, retain_e);
                    v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
                }
                st4(m.yin0 + (size_t)row * IN + 4 * c4, v);
            }
        }
        float2* d = reinterpret_cast<float2*>(sY + i * ldy + 4 * c4);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    for (int e = tid; e < GRU_ROWS * (D >> 2); e += GRU_NW * 64) {
        const int i = e / (D >> 2), c4 = e - i * (D >> 2), row = r0 + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) v = ld4(Hcur + (size_t)row * D + 4 * c4);
        float2* d = reinterpret_cast<float2*>(sH + i * ldh + 4 * c4);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    G4R_TICK(m, 0, 1);
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wh = m.dense_p + m.offWh[l];
    const GAS float* Wrz = m.dense_p + m.offWrz[l];
    const GAS float* Bh = m.dense_p + m.offBh[l];
    // ---- phase 1: V = y Wx (+ Bh) ; columns >= D additionally get H Wrz      (gru4rec.py:472-473)
    const int nct = (D3 + 15) >> 4;
    f32x4 acc[T1];
    long long bx[T1], bh[T1];
    unsigned mx = 0, mh = 0;
#pragma unroll
    for (int ti = 0; ti < T1; ++ti) {
        acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ct = wid + ti * GRU_NW, col = ct * 16 + li;
        bx[ti] = col < D3 ? col : -1;
        bh[ti] = (col >= D && col < D3) ? col - D : -1;
        if (ct < nct) { mx |= 1u << ti; if (ct * 16 + 15 >= D) mh |= 1u << ti; }
    }
    tile_gemm_rows<T1, U1>(acc, mx, bx, sY, ldy, IN, Wx, D3, li, lg);
    tile_gemm_rows<T1, U1>(acc, mh, bh, sH, ldh, D, Wrz, 2 * D, li, lg);
    __syncthreads();   // every wave is done reading sY before sV (same memory) is written
    G4R_TICK(m, 0, 2);
#pragma unroll
    for (int ti = 0; ti < T1; ++ti) {
        if ((mx >> ti) & 1u) {
            const int col = (wid + ti * GRU_NW) * 16 + li;
            if (col < D3) {
                const float bias = Bh[col];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) sV[(4 * lg + rg) * ldv + col] = acc[ti][rg] + bias;
            }
        }
    }
    __syncthreads();
    G4R_TICK(m, 0, 3);
    // ---- gates: r, z = sigmoid ; sH <- H*r ; z kept in sV
    for (int e = tid; e < GRU_ROWS * D; e += GRU_NW * 64) {
        const int i = e / D, d = e - i * D, row = r0 + i;
        const float rr = sigmoidf_(sV[i * ldv + D + d]);
        const float zz = sigmoidf_(sV[i * ldv + 2 * D + d]);
        const float hr = sH[i * ldh + d] * rr;
        sH[i * ldh + d] = hr;
        sV[i * ldv + 2 * D + d] = zz;
        if (train && row < M) {
            m.r[l][(size_t)row * D + d] = rr;
            m.z[l][(size_t)row * D + d] = zz;
            m.Hr[l][(size_t)row * D + d] = hr;
        }
    }
    __syncthreads();
    G4R_TICK(m, 0, 4);
    // ---- phase 2: c = act((H*r) Wh + V_c) ; h = (1-z) H + z c ; dropout ; reset   (gru4rec.py:474-479)
    const int nct2 = (D + 15) >> 4;
    f32x4 acc2[T2];
    long long b2[T2];
    unsigned m2 = 0;
#pragma unroll
    for (int ti = 0; ti < T2; ++ti) {
        acc2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ct = wid + ti * GRU_NW, col = ct * 16 + li;
        b2[ti] = col < D ? col : -1;
        if (ct < nct2) m2 |= 1u << ti;
    }
    tile_gemm_rows<T2, U2>(acc2, m2, b2, sH, ldh, D, Wh, D, li, lg);
    const GAS unsigned char* rst = train ? m.reset + t * m.B : nullptr;
#pragma unroll
    for (int ti = 0; ti < T2; ++ti) {
        if ((m2 >> ti) & 1u) {
            const int col = (wid + ti * GRU_NW) * 16 + li;
            if (col < D) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int i = 4 * lg + rg, row = r0 + i;
                    if (row < M) {
                        const float apre = acc2[ti][rg] + sV[i * ldv + col];
                        const float cc = act_fwd(m.hidden_act, m.ha_p0, m.ha_p1, apre);
                        const float zz = sV[i * ldv + 2 * D + col];
                        const float hv = Hcur[(size_t)row * D + col];
                        float h = (1.0f - zz) * hv + zz * cc;
                        if (train) {
                            if (m.drop_h > 0.f)
                                h *= drop_mult(m.seed, (unsigned)g, G4R_STREAM_DROP_HIDDEN + l, row, col, retain_h);
                            m.c[l][(size_t)row * D + col] = cc;
                            m.hd[l][(size_t)row * D + col] = h;
                            Hnext[(size_t)row * D + col] = rst[row] ? 0.f : h;
                        } else {
                            pa.hout[(size_t)row * D + col] = h;
                            Hnext[(size_t)row * D + col] = h;
                        }
                    }
                }
            }
        }
    }
    G4R_TICK(m, 0, 5);
}

// ---------------------------------------------------------------------------------------------
// Scoring GEMM: Sc[B, N] = h[B, D] * Wy[items]^T + By[items] - logq * lq[items]    (gru4rec.py:493-495)
// 128 rows x TN columns per workgroup; h tile and the gathered Wy rows are staged through LDS in
// K-chunks of <=128; fp32 MFMA 16x16x4.  Also publishes the column -> item map for the later kernels.
#define SC_BM 128
#define SC_KC 128
template <int TN>
__global__ __launch_bounds__(256) void k_score_fwd(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const StepCtx c = load_ctx(m);
    const int M = c.M, B = m.B, D = m.Dtop, N = m.N;
    const int ldk = SC_KC + 2;
    float* sA = smem;                      // [128][ldk]
    float* sB = sA + SC_BM * ldk;          // [TN][ldk]
    int* sItem = reinterpret_cast<int*>(sB + TN * ldk);   // [TN]
    const int n0 = blockIdx.x * TN, rbase = blockIdx.y * SC_BM;
    if (tid < TN) {
        const int n = n0 + tid;
        int item = -1;
        if (n < M) item = m.out_idx[c.t * B + n];
        else if (n >= B && n < N) item = m.ST[(size_t)(c.g % m.gl) * m.ns + (n - B)];
        sItem[tid] = item;
        if (blockIdx.y == 0 && n < m.ldSc) {
            m.col_item[n] = item;
            if (n < N) m.occ_idx[B + n] = item;
        }
    }
    if (rbase >= M) return;
    __syncthreads();
    constexpr int CT = TN / 16;
    f32x4 acc[2][CT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const GAS float* hsrc = m.hd[m.n_layers - 1];
    for (int kc0 = 0; kc0 < D; kc0 += SC_KC) {
        const int kc = min(SC_KC, D - kc0), kc4 = kc >> 2;
        for (int e = tid; e < SC_BM * kc4; e += 256) {
            const int i = e / kc4, c4 = e - i * kc4, row = rbase + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) v = ld4(hsrc + (size_t)row * D + kc0 + 4 * c4);
            float2* d = reinterpret_cast<float2*>(sA + i * ldk + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
        for (int e = tid; e < TN * kc4; e += 256) {
            const int j = e / kc4, c4 = e - j * kc4, item = sItem[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item >= 0) v = ld4(m.Wy + (size_t)item * D + kc0 + 4 * c4);
            float2* d = reinterpret_cast<float2*>(sB + j * ldk + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
        __syncthreads();
        for (int k = 0; k < kc; k += 4) {
            const float a0 = sA[(32 * wid + li) * ldk + k + lg];
            const float a1 = sA[(32 * wid + 16 + li) * ldk + k + lg];
#pragma unroll
            for (int cj = 0; cj < CT; ++cj) {
                const float b = sB[(16 * cj + li) * ldk + k + lg];
                acc[0][cj] = mfma16(a0, b, acc[0][cj]);
                acc[1][cj] = mfma16(a1, b, acc[1][cj]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int cj = 0; cj < CT; ++cj) {
        const int n = n0 + 16 * cj + li;
        const int item = sItem[16 * cj + li];
        float add = 0.f;
        if (item >= 0) {
            add = m.By[item];
            if (m.logq != 0.f) add -= m.logq * (n < B ? m.lq_tgt[item] : m.lq_smp[item]);
        }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = rbase + 32 * wid + 16 * ri + 4 * lg + rg;
                if (row < M && n < N) m.Sc[(size_t)row * m.ldSc + n] = acc[ri][cj][rg] + add;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-row final activation, loss and d cost / d s, in place in Sc.  One 256-thread workgroup per
// batch row; the row (N <= ~40K floats) is staged in LDS; row statistics via wave64 shuffles.
// Column j is active iff j < M (in-batch targets) or j >= B (sampled negatives); row i's positive is
// column i.  Losses: gru4rec.py:225-230 (cross_entropy), :239-241 (bpr_max), :245-248 (top1_max),
// softmax_neg :199-203.  The gradient goes through the softmax weights, as T.grad does.
__global__ __launch_bounds__(256) void k_loss_rows(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const StepCtx c = load_ctx(m);
    const int M = c.M, B = m.B, N = m.N, i = blockIdx.x;
    if (i >= M) return;
    float* sy = smem;              // [ldSc] yhat, later d/ds
    float* red = smem + m.ldSc;    // [8]
    GAS float* row = m.Sc + (size_t)i * m.ldSc;
#define ACTIVE(j) ((j) < M || (j) >= B)
    // ---- final activation (gru4rec.py:496)
    if (m.final_act == G4R_ACT_SOFTMAX) {
        float mx = -INFINITY;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) { const float v = row[j]; sy[j] = v; mx = fmaxf(mx, v); }
        mx = block_max_256(mx, red);
        float sm = 0.f;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) { const float e = fexp(sy[j] - mx); sy[j] = e; sm += e; }
        sm = block_sum_256(sm, red);
        const float inv_z = 1.f / sm;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) sy[j] = sy[j] * inv_z;
    } else {
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) sy[j] = act_fwd(m.final_act, m.fa_p0, m.fa_p1, row[j]);
    }
    __syncthreads();
    const float yd = sy[i];
    float Lrow = 0.f;
    // ---- loss and d L / d yhat (kept in registers per strided element, written back to sy)
    if (m.loss == G4R_LOSS_XE) {
        Lrow = -logf(yd + G4R_EPS_LOSS);
        __syncthreads();
        if (m.final_act == G4R_ACT_SOFTMAX) {
            // ds_k = yhat_k * (dy_k - sum_j dy_j yhat_j) with dy = -delta_ik / (yd + eps)
            const float coef = yd / (yd + G4R_EPS_LOSS);
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j)) sy[j] = coef * (sy[j] - (j == i ? 1.f : 0.f)) * m.inv_B;
        } else {
            const float dyd = -1.f / (yd + G4R_EPS_LOSS);
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j))
                    sy[j] = (j == i) ? dyd * act_bwd_from_out(m.final_act, m.fa_p0, m.fa_p1, yd) * m.inv_B : 0.f;
        }
    } else {
        // softmax over the negatives, with the positive zeroed first (so the max includes a 0)
        float mx = 0.f;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j) && j != i) mx = fmaxf(mx, sy[j]);
        mx = block_max_256(mx, red);
        float sm = 0.f;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j) && j != i) sm += fexp(sy[j] - mx);
        sm = block_sum_256(sm, red);
        const float inv_sm = 1.f / sm;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (m.loss == G4R_LOSS_BPR_MAX) {
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j) && j != i) {
                    const float y = sy[j], p = fexp(y - mx) * inv_sm, sg = sigmoidf_(yd - y);
                    s1 += sg * p;                 // A
                    s2 += y * y * p;              // Q
                    s3 += sg * (1.f - sg) * p;    // sum sigma' p
                }
        } else {
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j) && j != i) {
                    const float y = sy[j], p = fexp(y - mx) * inv_sm, u = sigmoidf_(y - yd), q = sigmoidf_(y * y);
                    s1 += p * (u + q);            // T
                    s3 += p * u * (1.f - u);
                }
        }
        s1 = block_sum_256(s1, red);
        s2 = block_sum_256(s2, red);
        s3 = block_sum_256(s3, red);
        float dyd;
        const float inv_A = 1.f / (s1 + G4R_EPS_LOSS);
        if (m.loss == G4R_LOSS_BPR_MAX) {
            Lrow = -logf(s1 + G4R_EPS_LOSS) + m.bpreg * s2;
            dyd = -s3 * inv_A;
        } else {
            Lrow = s1;
            dyd = -s3;
        }
        // d L / d yhat_j, written over yhat_j (the softmax final-act branch needs yhat again: keep a copy in row[])
        const bool fsm = (m.final_act == G4R_ACT_SOFTMAX);
        float inner = 0.f;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) {
                const float y = sy[j];
                float d;
                if (j == i) d = dyd;
                else {
                    const float p = fexp(y - mx) * inv_sm;
                    if (m.loss == G4R_LOSS_BPR_MAX) {
                        const float sg = sigmoidf_(yd - y);
                        d = -p * (sg - sg * (1.f - sg) - s1) * inv_A + m.bpreg * p * (2.f * y + y * y - s2);
                    } else {
                        const float u = sigmoidf_(y - yd), q = sigmoidf_(y * y);
                        d = p * (u + q - s1) + p * (u * (1.f - u) + 2.f * y * q * (1.f - q));
                    }
                }
                if (fsm) { row[j] = y; inner += d * y; sy[j] = d; }
                else sy[j] = d * act_bwd_from_out(m.final_act, m.fa_p0, m.fa_p1, y) * m.inv_B;
            }
        if (fsm) {
            inner = block_sum_256(inner, red);
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j)) { const float y = row[j]; sy[j] = y * (sy[j] - inner) * m.inv_B; }
        }
    }
    __syncthreads();
    for (int j = tid; j < m.ldSc; j += 256) row[j] = (j < N && ACTIVE(j)) ? sy[j] : 0.f;
    if (tid == 0) m.lossrow[i] = Lrow;
#undef ACTIVE
}

// ---------------------------------------------------------------------------------------------
// Scoring backward.  Role A (blockIdx.x < nblkA): dSy[N, D] = ds^T h and dSBy = colsum(ds); one wave
// per (16 columns n) x (up to 4 tiles of d).  Role B: split-K partials of dh = ds * Sy: one wave per
// (16 rows) x (<=4 tiles of d) x K-chunk, float4 reads of ds along n (K-permuted MFMA operands).
// All operands come straight from L2 (the whole step working set is L2/MALL resident).
#define SB_DG 4
__global__ __launch_bounds__(256) void k_score_bwd(const DevModel* __restrict__ mp, int nwavesA, int nblkA) {
    const DevModel& m = *mp;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const StepCtx c = load_ctx(m);
    const int M = c.M, B = m.B, D = m.Dtop, N = m.N, ld = m.ldSc;
    const int ndt = (D + 15) >> 4, ndg = (ndt + SB_DG - 1) / SB_DG;
    const GAS float* h = m.hd[m.n_layers - 1];
    if ((int)blockIdx.x < nblkA) {
        const int w = blockIdx.x * 4 + wid;
        if (w >= nwavesA) return;
        const int nt = w / ndg, dg = w - nt * ndg;
        const int n = nt * 16 + li;
        f32x4 acc[SB_DG];
#pragma unroll
        for (int q = 0; q < SB_DG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float asum = 0.f;
        constexpr int UA = 8;     // (1 + SB_DG) * UA independent loads per lane ahead of each MFMA batch
        for (int k0 = 0; k0 < M; k0 += 4 * UA) {
            float av[UA], bv[UA][SB_DG];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int b = k0 + 4 * u + lg;
                av[u] = (b < M) ? m.Sc[(size_t)b * ld + n] : 0.f;   // n < ldSc always (padded, zero-filled)
#pragma unroll
                for (int q = 0; q < SB_DG; ++q) {
                    const int d = (dg * SB_DG + q) * 16 + li;
                    bv[u][q] = (b < M && d < D) ? h[(size_t)b * D + d] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                asum += av[u];
#pragma unroll
                for (int q = 0; q < SB_DG; ++q)
                    if (dg * SB_DG + q < ndt) acc[q] = mfma16(av[u], bv[u][q], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < SB_DG; ++q) {
            const int dt = dg * SB_DG + q;
            if (dt < ndt) {
                const int d = dt * 16 + li;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nn = nt * 16 + 4 * lg + rg;
                    if (nn < N && d < D) m.dSy[(size_t)nn * D + d] = acc[q][rg];
                }
            }
        }
        if (dg == 0) {
            asum += __shfl_xor(asum, 16, 64);
            asum += __shfl_xor(asum, 32, 64);
            if (lg == 0 && n < N) m.dSBy[n] = asum;
        }
        return;
    }
    // ---- role B
    const int w = (blockIdx.x - nblkA) * 4 + wid;
    const int nrt = (B + 15) >> 4;
    const int per_kc = nrt * ndg;
    const int kc = w / per_kc;
    if (kc >= m.ksplit) return;
    const int rem = w - kc * per_kc, rt = rem / ndg, dg = rem - rt * ndg;
    const int r0 = rt * 16;
    if (r0 >= M) return;
    const int kbeg = kc * m.kch, kend = min(kbeg + m.kch, ld);
    f32x4 acc[SB_DG];
#pragma unroll
    for (int q = 0; q < SB_DG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int row = r0 + li;
    // ds (float4 along n) and the column -> item map of the whole K-chunk first, then per 16-wide k-step
    // all 4 * SB_DG gathered Wy operands before the MFMAs that consume them
    constexpr int MAXIT = 4;
    for (int kb = kbeg; kb < kend; kb += 16 * MAXIT) {
        float4 a4[MAXIT];
        int4 it[MAXIT];
#pragma unroll
        for (int s2 = 0; s2 < MAXIT; ++s2) {
            const int k0 = kb + 16 * s2 + 4 * lg;
            a4[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            it[s2] = make_int4(-1, -1, -1, -1);
            if (k0 < kend) {
                if (row < M) a4[s2] = ld4(m.Sc + (size_t)row * ld + k0);
                it[s2] = ldi4(m.col_item + k0);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < MAXIT; ++s2) {
            if (kb + 16 * s2 < kend) {      // wave-uniform
                const float av[4] = {a4[s2].x, a4[s2].y, a4[s2].z, a4[s2].w};
                const int iv[4] = {it[s2].x, it[s2].y, it[s2].z, it[s2].w};
                float bv[4][SB_DG];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < SB_DG; ++q) {
                        const int d = (dg * SB_DG + q) * 16 + li;
                        bv[e][q] = (iv[e] >= 0 && d < D) ? m.Wy[(size_t)iv[e] * D + d] : 0.f;
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < SB_DG; ++q)
                        if (dg * SB_DG + q < ndt) acc[q] = mfma16(av[e], bv[e][q], acc[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SB_DG; ++q) {
        const int dt = dg * SB_DG + q;
        if (dt < ndt) {
            const int d = dt * 16 + li;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int rr = r0 + 4 * lg + rg;
                if (rr < M && d < D) m.dhpart[((size_t)kc * B + rr) * D + d] = acc[q][rg];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GRU backward, row-local part (no BPTT: H is a constant input, gru4rec.py:460-463,576).
//   dz = dh (c - H) ; dc = dh z ; da = dc act'(c) ; dr = (da Wh^T) H ; d(pre-sigmoid) ; dV = [da | drp | dzp]
//   dy = dV Wx^T  -> embedding-row gradient dSx (layer 0) or the lower layer's dh.
template <int T2, int U2>
__global__ __launch_bounds__(GRU_NW * 64) void k_gru_bwd_rows(const DevModel* __restrict__ mp, int l) {
    const DevModel& m = *mp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const StepCtx c = load_ctx(m);
    const int M = c.M, B = m.B, D = m.D[l], IN = m.IN[l], D3 = 3 * D;
    const int ldv = D3 + 2;
    float* sDV = smem;   // [16][ldv]
    const int r0 = blockIdx.x * GRU_ROWS;
    if (r0 >= M) return;
    const GAS float* Hcur = m.H[l][c.g & 1];
    const bool top = (l == m.n_layers - 1);
    const float retain_h = 1.0f - m.drop_h, retain_e = 1.0f - m.drop_e;
    for (int e = tid; e < GRU_ROWS * D; e += GRU_NW * 64) {
        const int i = e / D, d = e - i * D, row = r0 + i;
        float da = 0.f, dzp = 0.f;
        if (row < M) {
            float dh;
            if (top) {
                dh = 0.f;
                const GAS float* pp = m.dhpart + (size_t)row * D + d;
                const size_t ps = (size_t)B * D;
                int kc = 0;
                for (; kc + 8 <= m.ksplit; kc += 8) {     // 8 independent loads in flight, fixed summation order
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = pp[(size_t)(kc + q) * ps];
#pragma unroll
                    for (int q = 0; q < 8; ++q) dh += v[q];
                }
                for (; kc < m.ksplit; ++kc) dh += pp[(size_t)kc * ps];
            } else {
                dh = m.dyl[l][(size_t)row * D + d];
            }
            if (m.drop_h > 0.f) dh *= drop_mult(m.seed, (unsigned)c.g, G4R_STREAM_DROP_HIDDEN + l, row, d, retain_h);
            const size_t o = (size_t)row * D + d;
            const float hv = Hcur[o], zz = m.z[l][o], cc = m.c[l][o];
            const float dz = dh * (cc - hv), dc = dh * zz;
            da = dc * act_bwd_from_out(m.hidden_act, m.ha_p0, m.ha_p1, cc);
            dzp = dz * zz * (1.f - zz);
        }
        sDV[i * ldv + d] = da;
        sDV[i * ldv + 2 * D + d] = dzp;
    }
    __syncthreads();
    const GAS float* Wx = m.dense_p + m.offWx[l];
    const GAS float* Wh = m.dense_p + m.offWh[l];
    // ---- dHr = da Wh^T ; drp = dHr * H * r (1 - r)
    {
        const int nct = (D + 15) >> 4;
        f32x4 acc[T2];
        long long bo[T2];
        unsigned tm = 0;
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int ct = wid + ti * GRU_NW, col = ct * 16 + li;
            bo[ti] = col < D ? (long long)col * D : -1;     // B[k][j] = Wh[j][k]
            if (ct < nct) tm |= 1u << ti;
        }
        tile_gemm_rows<T2, U2>(acc, tm, bo, sDV, ldv, D, Wh, 1, li, lg);
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            if ((tm >> ti) & 1u) {
                const int col = (wid + ti * GRU_NW) * 16 + li;
                if (col < D) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int i = 4 * lg + rg, row = r0 + i;
                        float drp = 0.f;
                        if (row < M) {
                            const size_t o = (size_t)row * D + col;
                            const float rr = m.r[l][o];
                            drp = acc[ti][rg] * Hcur[o] * rr * (1.f - rr);
                        }
                        sDV[i * ldv + D + col] = drp;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- publish dV rows for the batch-contraction kernel
    for (int e = tid; e < GRU_ROWS * D3; e += GRU_NW * 64) {
        const int i = e / D3, q = e - i * D3, row = r0 + i;
        if (row < M) m.dV[l][(size_t)row * D3 + q] = sDV[i * ldv + q];
    }
    // ---- dy = dV Wx^T
    {
        const int nct = (IN + 15) >> 4;
        f32x4 acc[T2];
        long long bo[T2];
        unsigned tm = 0;
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int ct = wid + ti * GRU_NW, col = ct * 16 + li;
            bo[ti] = col < IN ? (long long)col * D3 : -1;    // B[k][j] = Wx[j][k]
            if (ct < nct) tm |= 1u << ti;
        }
        tile_gemm_rows<T2, U2>(acc, tm, bo, sDV, ldv, D3, Wx, 1, li, lg);
#pragma unroll
        for (int ti = 0; ti < T2; ++ti) {
            if ((tm >> ti) & 1u) {
                const int col = (wid + ti * GRU_NW) * 16 + li;
                if (col < IN) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int row = r0 + 4 * lg + rg;
                        if (row < M) {
                            float v = acc[ti][rg];
                            if (l == 0) {
                                if (m.drop_e > 0.f)
                                    v *= drop_mult(m.seed, (unsigned)c.g, G4R_STREAM_DROP_EMBED, row, col, retain_e);
                                m.dSx[(size_t)row * IN + col] = v;
                            } else {
                                m.dyl[l - 1][(size_t)row * IN + col] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense gradients: contractions over the batch, one wave per 16x16 output tile of
//   dWx = yin^T dV ; dWh = (H r)^T dV[:, :D] ; dWrz = H^T dV[:, D:] ; dBh = colsum(dV)
// with the dense Adagrad(+momentum) update (gru4rec.py:330-334,390-406) fused into the epilogue when
// no all-reduce is needed (single GPU); otherwise the gradient goes to dense_g for RCCL.
// One 16x16 output tile of a dense GRU gradient, resolved on the host: out[r0.., c0..] (leading dim ldo, at
// float offset `base` of the flat dense buffers) = X^T[., batch] * dV[batch, coff + .] ; X0/X1 = operand for
// even/odd global step (the hidden state ping-pongs) ; X == nullptr selects the bias row (column sums of dV).
struct DenseTile {
    GP(const float) X0; GP(const float) X1; GP(const float) dV;
    long long base;
    int ldx, ldv, nrows, ncols, coff, ldo, r0, c0;
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

__global__ __launch_bounds__(256) void k_dense_grad(const DevModel* __restrict__ mp, const DenseTile* __restrict__ tiles_, int ntiles) {
    const DevModel& m = *mp;
    const GAS DenseTile* tiles = (const GAS DenseTile*)tiles_;   // same mangled signature on both passes
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x * 4 + wid;
    if (w >= ntiles) return;
    const StepCtx c = load_ctx(m);
    const DenseTile tl = tiles[w];            // fully resolved on the host: no per-layer lookups here
    const GAS float* X = (c.g & 1) ? tl.X1 : tl.X0;
    const GAS float* dV = tl.dV;
    const int M = c.M, ra = tl.r0 + li, cb = tl.c0 + li;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 16;       // 2*U independent loads in flight per lane before the MFMAs of a batch
    for (int k0 = 0; k0 < M; k0 += 4 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = k0 + 4 * u + lg;
            av[u] = 0.f; bv[u] = 0.f;
            if (b < M) {
                if (X == nullptr) av[u] = (li == 0) ? 1.f : 0.f;      // bias row: column sums of dV
                else if (ra < tl.nrows) av[u] = X[(size_t)b * tl.ldx + ra];
                if (cb < tl.ncols) bv[u] = dV[(size_t)b * tl.ldv + tl.coff + cb];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16(av[u], bv[u], acc);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int row = tl.r0 + 4 * lg + rg;
        if (row < tl.nrows && cb < tl.ncols) {
            const size_t off = (size_t)tl.base + (size_t)row * tl.ldo + cb;
            if (m.apply_dense_inplace) dense_adagrad(m, off, acc[rg]);
            else m.dense_g[off] = acc[rg];
        }
    }
}

// after the RCCL all-reduce: element-wise dense Adagrad on the averaged gradient
__global__ __launch_bounds__(256) void k_dense_apply(const DevModel* __restrict__ mp) {
    const DevModel& m = *mp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m.dense_count) dense_adagrad(m, (size_t)i, m.dense_g[i] * m.grad_scale);
}

// ---------------------------------------------------------------------------------------------
// Sparse Adagrad(+momentum) on the gathered rows, gru4rec.py:335-340,407-431, with the reference's
// duplicate-index semantics made deterministic:
//   - every occurrence is scaled with the PRE-step accumulator: g~ = g / sqrt(acc_old + g^2 + eps)
//   - parameter increments of duplicates accumulate, in occurrence order (inc_subtensor)
//   - accumulator / velocity take the value of the LAST occurrence (set_subtensor, NumPy order)
// One wave per occurrence k of (X | Y | samples).  The wave of the last occurrence of an item owns
// the row: it scans the occurrence list for its duplicates (ballot over 64 entries at a time) and
// applies them in ascending order.  No atomics, no scratch state, bit-reproducible.
// The extra last block folds the per-row losses into loss_steps[t] and advances the step state.
#define SP_WAVES 8   // occurrences (waves) per workgroup

// MAXCH = float4 chunks per lane (1: row width <= 256, 2: <= 512).  One wave per occurrence k of
// (X | Y | samples); the wave of an item's LAST occurrence owns the row and applies all of the item's
// occurrences in ascending order (semantics: comment block above).  The occurrence list is
// staged in LDS once per workgroup; the owner keeps its duplicate list in registers (entry i in lane i) and
// fetches the gradient rows of up to UB duplicates together, so a hot item costs cnt/UB memory round trips
// instead of cnt.  The extra last block folds the per-row losses into loss_steps[t] and advances the step state.
template <int MAXCH>
__global__ __launch_bounds__(SP_WAVES * 64) void k_sparse_update(const DevModel* __restrict__ mp, int nblk_occ) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    int* sOcc = reinterpret_cast<int*>(smem);     // occurrence list, padded with -2 to a multiple of 256 (+256)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const StepCtx c = load_ctx(m);
    const int B = m.B, R = m.R;
    if ((int)blockIdx.x == nblk_occ) {
        // ---- bookkeeping block: cost = sum_i L_i / batch_size (gru4rec.py:577), NaN flag (:626), advance state
        if (wid == 0) {
            float s = 0.f;
            for (int i = lane; i < c.M; i += 64) s += m.lossrow[i];
            s = wave_sum(s);
            if (lane == 0) {
                const float cost = s * m.inv_B;
                m.loss_steps[c.t] = cost;
                if (isnan(cost)) m.st->nan_flag = 1;
                m.st->t_a = c.t + 1;
                m.st->g_a = c.g + 1;
            }
        }
        return;
    }
    const long long t_start = m.dbgclk ? wall_clock64() : 0;
    G4R_TICK(m, 1, 0);
    const int Rpad = ((R + 255) & ~255) + 256;
    for (int j = tid; j < Rpad; j += SP_WAVES * 64) sOcc[j] = j < R ? m.occ_idx[j] : -2;
    __syncthreads();
    G4R_TICK(m, 1, 1);
    const int k = blockIdx.x * SP_WAVES + wid;
    if (k >= R) return;
    const int item = sOcc[k];
    if (item < 0) return;
    const bool constrained = (m.embed_mode == G4R_EMBED_CONSTRAINED);
    // occurrence range sharing a table with k: constrained -> all of X|Y|samples ; separate -> X alone, Y|samples alone
    const int lo = (constrained || k < B) ? 0 : B;
    const int hi = (constrained || k >= B) ? R : B;
    // ---- is there a later occurrence of the same item?  then that wave owns the row
    for (int base = (k + 1) & ~255; base < hi; base += 256) {
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sOcc[base + 64 * e + lane];
        bool later = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int j = base + 64 * e + lane; later |= (j > k && j < hi && v[e] == item); }
        if (__ballot(later)) return;
    }
    const long long t_own = m.dbgclk ? wall_clock64() : 0;
    G4R_TICK(m, 1, 2);
    // ---- row state (pre-step values; every occurrence is scaled with the pre-step accumulator)
    const bool tableE = (k < B && !constrained);
    GAS float* P = tableE ? m.E : m.Wy;
    GAS float* A = tableE ? m.accE : m.accWy;
    GAS float* V = tableE ? m.velE : m.velWy;
    const int W = tableE ? m.Ein : m.Dtop;
    const int nc4 = W >> 2;
    const bool mom = m.mom > 0.f;
    const bool bias = (k >= B);
    float pc[MAXCH][4], pz[MAXCH][4], az[MAXCH][4], vz[MAXCH][4], al[MAXCH][4], vl[MAXCH][4];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int c4 = lane + 64 * q;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), a0 = p0, v0 = p0;
        if (c4 < nc4) {
            const size_t o = (size_t)item * W + 4 * c4;
            p0 = ld4(P + o);
            a0 = ld4(A + o);
            if (mom) v0 = ld4(V + o);
        }
        pz[q][0] = p0.x; pz[q][1] = p0.y; pz[q][2] = p0.z; pz[q][3] = p0.w;
        az[q][0] = a0.x; az[q][1] = a0.y; az[q][2] = a0.z; az[q][3] = a0.w;
        vz[q][0] = v0.x; vz[q][1] = v0.y; vz[q][2] = v0.z; vz[q][3] = v0.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pc[q][e] = pz[q][e]; al[q][e] = az[q][e]; vl[q][e] = vz[q][e]; }
    }
    // output bias By: occurrences among Y|samples only (gru4rec.py:486-489)
    float bp = 0.f, bpz = 0.f, baz = 0.f, bvz = 0.f, bal = 0.f, bvl = 0.f;
    if (bias) {
        bp = m.By[item]; bpz = bp; baz = m.accBy[item]; bal = baz;
        if (mom) { bvz = m.velBy[item]; bvl = bvz; }
    }
    constexpr int UB = (MAXCH == 1) ? 16 : 8;
    // apply the duplicates listed one-per-lane in myj[0..cnt), ascending occurrence order.  Branch-free per batch:
    // the UB x 4 x MAXCH scale factors of a batch are independent, only the parameter subtraction is a chain.
    auto apply = [&](int myj, int cnt) {
        const int jb = (lane < cnt && myj >= B) ? myj : -1;
        float dlt_b = 0.f, an_b = 0.f;     // lane i: bias update of duplicate i
        if (bias) {
            const float gb = (jb >= 0) ? m.dSBy[jb - B] : 0.f;     // all bias gradients of the list in one round
            an_b = baz + gb * gb;
            const float gs = gb * frsq(an_b + G4R_EPS_ADAGRAD);
            dlt_b = (m.lmbd > 0.f) ? m.lr * (gs + m.lmbd * bpz) : m.lr * gs;
        }
        for (int i0 = 0; i0 < cnt; i0 += UB) {
            float4 g[UB][MAXCH];
#pragma unroll
            for (int u = 0; u < UB; ++u) {          // gradient rows of up to UB duplicates in flight together
                const int jj = __builtin_amdgcn_readlane(myj, (i0 + u) & 63);
#pragma unroll
                for (int q = 0; q < MAXCH; ++q) {
                    const int c4 = lane + 64 * q;
                    g[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i0 + u < cnt && c4 < nc4) {
                        const GAS float* grow = (jj < B) ? m.dSx + (size_t)jj * W : m.dSy + (size_t)(jj - B) * W;
                        g[u][q] = ld4(grow + 4 * c4);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const bool valid = (i0 + u < cnt);
#pragma unroll
                for (int q = 0; q < MAXCH; ++q) {
                    const float gv[4] = {g[u][q].x, g[u][q].y, g[u][q].z, g[u][q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float an = az[q][e] + gv[e] * gv[e];
                        const float gs = gv[e] * frsq(an + G4R_EPS_ADAGRAD);
                        float delta = (m.lmbd > 0.f) ? m.lr * (gs + m.lmbd * pz[q][e]) : m.lr * gs;
                        delta = valid ? delta : 0.f;
                        al[q][e] = valid ? an : al[q][e];
                        if (mom) {
                            const float v2 = m.mom * vz[q][e] - delta;
                            vl[q][e] = valid ? v2 : vl[q][e];
                            pc[q][e] = pc[q][e] + (valid ? v2 : 0.f);
                        } else {
                            pc[q][e] = pc[q][e] - delta;
                        }
                    }
                }
            }
        }
        if (bias) {
            for (int i = 0; i < cnt; ++i) {          // ordered chain over scalar (SGPR) broadcasts
                const int jj = __builtin_amdgcn_readlane(jb, i);
                if (jj < 0) continue;
                const float delta = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlt_b), i));
                bal = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, an_b), i));
                if (mom) { const float v2 = m.mom * bvz - delta; bvl = v2; bp = bp + v2; }
                else bp = bp - delta;
            }
        }
    };
    // ---- collect this item's occurrences in [lo, k] (ascending) 64 at a time and apply them
    int myj = -1, cnt = 0, total = 0;
    for (int base = lo & ~255; base <= k; base += 256) {
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sOcc[base + 64 * e + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = base + 64 * e + lane;
            unsigned long long mask = __ballot(j >= lo && j <= k && v[e] == item);
            while (mask) {
                const int bit = __ffsll((unsigned long long)mask) - 1;
                mask &= mask - 1;
                if (lane == cnt) myj = base + 64 * e + bit;
                ++total;
                if (++cnt == 64) { apply(myj, 64); cnt = 0; myj = -1; }
            }
        }
    }
    const long long t_col = m.dbgclk ? wall_clock64() : 0;
    G4R_TICK(m, 1, 3);
    if (cnt) apply(myj, cnt);
    const long long t_app = m.dbgclk ? wall_clock64() : 0;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < nc4) {
            const size_t o = (size_t)item * W + 4 * c4;
            st4(P + o, make_float4(pc[q][0], pc[q][1], pc[q][2], pc[q][3]));
            st4(A + o, make_float4(al[q][0], al[q][1], al[q][2], al[q][3]));
            if (mom) st4(V + o, make_float4(vl[q][0], vl[q][1], vl[q][2], vl[q][3]));
        }
    }
    if (bias && lane == 0) {
        m.By[item] = bp;
        m.accBy[item] = bal;
        if (mom) m.velBy[item] = bvl;
    }
    G4R_TICK(m, 1, 4);
    if (m.dbgclk && lane == 0) {
        const long long t_end = wall_clock64();
        const long long dur = t_end - t_start;
        GAS long long* tr = m.dbgclk + 64 + 8 * k;
        tr[0] = t_start; tr[1] = t_own; tr[2] = t_col; tr[3] = t_app; tr[4] = t_end; tr[5] = total; tr[6] = 0; tr[7] = item;
        (void)dur;
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

__global__ void k_set_state(StepState* st, long long t, long long g) {
    st->t_a = t; st->t_b = t; st->g_a = g; st->g_b = g;
}
