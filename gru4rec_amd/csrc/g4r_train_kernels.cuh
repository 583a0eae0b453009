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
#define GRU_MAXT 12       // max 16-col tiles per wave in phase 1  (=> 3*D <= 16*12*8)
#define GRU_MAXT2 4       // max tiles per wave for D- or IN-wide outputs (=> D, IN <= 512)

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
    const int* in_idx;   // device, layer 0 gather indices
    const float* ysrc;   // layer > 0 input rows
    const float* Hcur;
    float* Hnext;
    float* hout;         // [rows][D]
    int M;
};

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
    const float *Hcur, *ysrc = nullptr;
    float* Hnext;
    const int* gidx = nullptr;
    if (train) {
        StepCtx c = first ? load_ctx_first(m) : load_ctx(m);
        t = c.t; g = c.g; M = c.M;
        Hcur = m.H[l][g & 1];
        Hnext = m.H[l][(g + 1) & 1];
        if (l == 0) gidx = m.in_idx + t * m.B; else ysrc = m.hd[l - 1];
    } else {
        M = pa.M; Hcur = pa.Hcur; Hnext = pa.Hnext; gidx = pa.in_idx; ysrc = pa.ysrc;
    }
    const int r0 = blockIdx.x * GRU_ROWS;
    if (train && l == 0 && tid < GRU_ROWS) {
        const int row = r0 + tid;
        if (row < m.B) m.occ_idx[row] = row < M ? gidx[row] : -1;
    }
    if (r0 >= M) return;
    const float* table = (m.embed_mode == G4R_EMBED_CONSTRAINED) ? m.Wy : m.E;
    const float retain_e = 1.0f - m.drop_e, retain_h = 1.0f - m.drop_h;
    // ---- stage input rows (gather + embedding dropout) and hidden rows
    for (int e = tid; e < GRU_ROWS * (IN >> 2); e += GRU_NW * 64) {
        const int i = e / (IN >> 2), c4 = e - i * (IN >> 2), row = r0 + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) {
            const float* src = (l == 0) ? table + (size_t)gidx[row] * IN : ysrc + (size_t)row * IN;
            v = *reinterpret_cast<const float4*>(src + 4 * c4);
            if (train && l == 0) {
                if (m.drop_e > 0.f) {
                    const float4 mk = drop_mult4(m.seed, (unsigned)g, G4R_STREAM_DROP_EMBED, row, c4, retain_e);
                    v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
                }
                *reinterpret_cast<float4*>(m.yin0 + (size_t)row * IN + 4 * c4) = v;
            }
        }
        float2* d = reinterpret_cast<float2*>(sY + i * ldy + 4 * c4);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    for (int e = tid; e < GRU_ROWS * (D >> 2); e += GRU_NW * 64) {
        const int i = e / (D >> 2), c4 = e - i * (D >> 2), row = r0 + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) v = *reinterpret_cast<const float4*>(Hcur + (size_t)row * D + 4 * c4);
        float2* d = reinterpret_cast<float2*>(sH + i * ldh + 4 * c4);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    const float* Wx = m.dense_p + m.offWx[l];
    const float* Wh = m.dense_p + m.offWh[l];
    const float* Wrz = m.dense_p + m.offWrz[l];
    const float* Bh = m.dense_p + m.offBh[l];
    // ---- phase 1: V = y Wx (+ Bh) ; columns >= D additionally get H Wrz      (gru4rec.py:472-473)
    const int nct = (D3 + 15) >> 4;
    int ntw = 0;
    for (int ct = wid; ct < nct; ct += GRU_NW) ++ntw;
    f32x4 acc[GRU_MAXT];
#pragma unroll
    for (int ti = 0; ti < GRU_MAXT; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < IN; k += 4) {
        const float a = sY[li * ldy + k + lg];
        const float* wrow = Wx + (size_t)(k + lg) * D3;
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT; ++ti) {
            if (ti < ntw) {
                const int col = (wid + ti * GRU_NW) * 16 + li;
                const float b = col < D3 ? wrow[col] : 0.f;
                acc[ti] = mfma16(a, b, acc[ti]);
            }
        }
    }
    for (int k = 0; k < D; k += 4) {
        const float a = sH[li * ldh + k + lg];
        const float* wrow = Wrz + (size_t)(k + lg) * (2 * D);
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT; ++ti) {
            if (ti < ntw) {
                const int c0 = (wid + ti * GRU_NW) * 16;
                if (c0 + 15 >= D) {   // wave-uniform: tile touches the r/z column blocks
                    const int col = c0 + li;
                    const float b = (col >= D && col < D3) ? wrow[col - D] : 0.f;
                    acc[ti] = mfma16(a, b, acc[ti]);
                }
            }
        }
    }
    __syncthreads();   // every wave is done reading sY before sV (same memory) is written
#pragma unroll
    for (int ti = 0; ti < GRU_MAXT; ++ti) {
        if (ti < ntw) {
            const int col = (wid + ti * GRU_NW) * 16 + li;
            if (col < D3) {
                const float bias = Bh[col];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) sV[(4 * lg + rg) * ldv + col] = acc[ti][rg] + bias;
            }
        }
    }
    __syncthreads();
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
    // ---- phase 2: c = act((H*r) Wh + V_c) ; h = (1-z) H + z c ; dropout ; reset   (gru4rec.py:474-479)
    const int nct2 = (D + 15) >> 4;
    int ntw2 = 0;
    for (int ct = wid; ct < nct2; ct += GRU_NW) ++ntw2;
    f32x4 acc2[GRU_MAXT2];
#pragma unroll
    for (int ti = 0; ti < GRU_MAXT2; ++ti) acc2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; k += 4) {
        const float a = sH[li * ldh + k + lg];
        const float* wrow = Wh + (size_t)(k + lg) * D;
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT2; ++ti) {
            if (ti < ntw2) {
                const int col = (wid + ti * GRU_NW) * 16 + li;
                const float b = col < D ? wrow[col] : 0.f;
                acc2[ti] = mfma16(a, b, acc2[ti]);
            }
        }
    }
    const unsigned char* rst = train ? m.reset + t * m.B : nullptr;
#pragma unroll
    for (int ti = 0; ti < GRU_MAXT2; ++ti) {
        if (ti < ntw2) {
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
    const float* hsrc = m.hd[m.n_layers - 1];
    for (int kc0 = 0; kc0 < D; kc0 += SC_KC) {
        const int kc = min(SC_KC, D - kc0), kc4 = kc >> 2;
        for (int e = tid; e < SC_BM * kc4; e += 256) {
            const int i = e / kc4, c4 = e - i * kc4, row = rbase + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) v = *reinterpret_cast<const float4*>(hsrc + (size_t)row * D + kc0 + 4 * c4);
            float2* d = reinterpret_cast<float2*>(sA + i * ldk + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
        for (int e = tid; e < TN * kc4; e += 256) {
            const int j = e / kc4, c4 = e - j * kc4, item = sItem[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item >= 0) v = *reinterpret_cast<const float4*>(m.Wy + (size_t)item * D + kc0 + 4 * c4);
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
    float* row = m.Sc + (size_t)i * m.ldSc;
#define ACTIVE(j) ((j) < M || (j) >= B)
    // ---- final activation (gru4rec.py:496)
    if (m.final_act == G4R_ACT_SOFTMAX) {
        float mx = -INFINITY;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) { const float v = row[j]; sy[j] = v; mx = fmaxf(mx, v); }
        mx = block_max_256(mx, red);
        float sm = 0.f;
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) { const float e = expf(sy[j] - mx); sy[j] = e; sm += e; }
        sm = block_sum_256(sm, red);
        for (int j = tid; j < N; j += 256)
            if (ACTIVE(j)) sy[j] = sy[j] / sm;
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
            if (ACTIVE(j) && j != i) sm += expf(sy[j] - mx);
        sm = block_sum_256(sm, red);
        const float inv_sm = 1.f / sm;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (m.loss == G4R_LOSS_BPR_MAX) {
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j) && j != i) {
                    const float y = sy[j], p = expf(y - mx) * inv_sm, sg = sigmoidf_(yd - y);
                    s1 += sg * p;                 // A
                    s2 += y * y * p;              // Q
                    s3 += sg * (1.f - sg) * p;    // sum sigma' p
                }
        } else {
            for (int j = tid; j < N; j += 256)
                if (ACTIVE(j) && j != i) {
                    const float y = sy[j], p = expf(y - mx) * inv_sm, u = sigmoidf_(y - yd), q = sigmoidf_(y * y);
                    s1 += p * (u + q);            // T
                    s3 += p * u * (1.f - u);
                }
        }
        s1 = block_sum_256(s1, red);
        s2 = block_sum_256(s2, red);
        s3 = block_sum_256(s3, red);
        float dyd;
        if (m.loss == G4R_LOSS_BPR_MAX) {
            Lrow = -logf(s1 + G4R_EPS_LOSS) + m.bpreg * s2;
            dyd = -s3 / (s1 + G4R_EPS_LOSS);
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
                    const float p = expf(y - mx) * inv_sm;
                    if (m.loss == G4R_LOSS_BPR_MAX) {
                        const float sg = sigmoidf_(yd - y);
                        d = -p * (sg - sg * (1.f - sg) - s1) / (s1 + G4R_EPS_LOSS) + m.bpreg * p * (2.f * y + y * y - s2);
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
    const float* h = m.hd[m.n_layers - 1];
    if ((int)blockIdx.x < nblkA) {
        const int w = blockIdx.x * 4 + wid;
        if (w >= nwavesA) return;
        const int nt = w / ndg, dg = w - nt * ndg;
        const int n = nt * 16 + li;
        f32x4 acc[SB_DG];
#pragma unroll
        for (int q = 0; q < SB_DG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float asum = 0.f;
        for (int k = 0; k < M; k += 4) {
            const int b = k + lg;
            const float a = (b < M) ? m.Sc[(size_t)b * ld + n] : 0.f;   // n < ldSc always (padded, zero-filled)
            asum += a;
#pragma unroll
            for (int q = 0; q < SB_DG; ++q) {
                const int dt = dg * SB_DG + q;
                if (dt < ndt) {
                    const int d = dt * 16 + li;
                    const float bv = (b < M && d < D) ? h[(size_t)b * D + d] : 0.f;
                    acc[q] = mfma16(a, bv, acc[q]);
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
    for (int k0 = kbeg + 4 * lg; k0 < kend + 4 * lg; k0 += 16) {   // all 4 lane groups run the same trip count
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 it = make_int4(-1, -1, -1, -1);
        if (k0 < kend) {
            if (row < M) a4 = *reinterpret_cast<const float4*>(m.Sc + (size_t)row * ld + k0);
            it = *reinterpret_cast<const int4*>(m.col_item + k0);
        }
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        const int iv[4] = {it.x, it.y, it.z, it.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int q = 0; q < SB_DG; ++q) {
                const int dt = dg * SB_DG + q;
                if (dt < ndt) {
                    const int d = dt * 16 + li;
                    const float bv = (iv[e] >= 0 && d < D) ? m.Wy[(size_t)iv[e] * D + d] : 0.f;
                    acc[q] = mfma16(av[e], bv, acc[q]);
                }
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
    const float* Hcur = m.H[l][c.g & 1];
    const bool top = (l == m.n_layers - 1);
    const float retain_h = 1.0f - m.drop_h, retain_e = 1.0f - m.drop_e;
    for (int e = tid; e < GRU_ROWS * D; e += GRU_NW * 64) {
        const int i = e / D, d = e - i * D, row = r0 + i;
        float da = 0.f, dzp = 0.f;
        if (row < M) {
            float dh;
            if (top) {
                dh = 0.f;
                for (int kc = 0; kc < m.ksplit; ++kc) dh += m.dhpart[((size_t)kc * B + row) * D + d];
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
    const float* Wx = m.dense_p + m.offWx[l];
    const float* Wh = m.dense_p + m.offWh[l];
    // ---- dHr = da Wh^T ; drp = dHr * H * r (1 - r)
    {
        const int nct = (D + 15) >> 4;
        int ntw = 0;
        for (int ct = wid; ct < nct; ct += GRU_NW) ++ntw;
        f32x4 acc[GRU_MAXT2];
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT2; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < D; k += 4) {
            const float a = sDV[li * ldv + k + lg];
#pragma unroll
            for (int ti = 0; ti < GRU_MAXT2; ++ti) {
                if (ti < ntw) {
                    const int col = (wid + ti * GRU_NW) * 16 + li;
                    const float b = col < D ? Wh[(size_t)col * D + k + lg] : 0.f;   // B[k][j] = Wh[j][k]
                    acc[ti] = mfma16(a, b, acc[ti]);
                }
            }
        }
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT2; ++ti) {
            if (ti < ntw) {
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
        int ntw = 0;
        for (int ct = wid; ct < nct; ct += GRU_NW) ++ntw;
        f32x4 acc[GRU_MAXT2];
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT2; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < D3; k += 4) {
            const float a = sDV[li * ldv + k + lg];
#pragma unroll
            for (int ti = 0; ti < GRU_MAXT2; ++ti) {
                if (ti < ntw) {
                    const int col = (wid + ti * GRU_NW) * 16 + li;
                    const float b = col < IN ? Wx[(size_t)col * D3 + k + lg] : 0.f;   // B[k][j] = Wx[j][k]
                    acc[ti] = mfma16(a, b, acc[ti]);
                }
            }
        }
#pragma unroll
        for (int ti = 0; ti < GRU_MAXT2; ++ti) {
            if (ti < ntw) {
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
    const float *X0, *X1, *dV;
    long long base;
    int ldx, ldv, nrows, ncols, coff, ldo, r0, c0;
};

__device__ __forceinline__ void dense_adagrad(const DevModel& m, size_t off, float g) {
    const float acc = m.dense_acc[off] + g * g;
    m.dense_acc[off] = acc;
    const float gs = g / sqrtf(acc + G4R_EPS_ADAGRAD);
    const float p = m.dense_p[off];
    if (m.mom > 0.f) {
        const float v = m.mom * m.dense_vel[off] - m.lr * (gs + m.lmbd * p);
        m.dense_vel[off] = v;
        m.dense_p[off] = p + v;
    } else {
        m.dense_p[off] = p * (1.0f - m.lr * m.lmbd) - m.lr * gs;
    }
}

__global__ __launch_bounds__(256) void k_dense_grad(const DevModel* __restrict__ mp, const DenseTile* __restrict__ tiles, int ntiles) {
    const DevModel& m = *mp;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x * 4 + wid;
    if (w >= ntiles) return;
    const StepCtx c = load_ctx(m);
    const DenseTile tl = tiles[w];            // fully resolved on the host: no per-layer lookups here
    const float* X = (c.g & 1) ? tl.X1 : tl.X0;
    const float* dV = tl.dV;
    const int M = c.M, ra = tl.r0 + li, cb = tl.c0 + li;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < M; k += 4) {
        const int b = k + lg;
        float a = 0.f, bv = 0.f;
        if (b < M) {
            if (X == nullptr) a = (li == 0) ? 1.f : 0.f;      // bias row: column sums of dV
            else if (ra < tl.nrows) a = X[(size_t)b * tl.ldx + ra];
            if (cb < tl.ncols) bv = dV[(size_t)b * tl.ldv + tl.coff + cb];
        }
        acc = mfma16(a, bv, acc);
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
#define SP_MAXCH 2   // float4 chunks per lane: row width <= 4*64*SP_MAXCH = 512
__device__ __forceinline__ void sparse_row_update(const DevModel& m, float* P, float* A, float* V, int item, int W,
                                                  const int* occ, int lo, int k, const float* gx, const float* gy,
                                                  int B, int lane) {
    // gradient row of occurrence j: j < B -> gx[j] (width W) ; else gy[j - B]
    const int nc4 = W >> 2;
    const bool mom = m.mom > 0.f;
    float pc[SP_MAXCH][4], pz[SP_MAXCH][4], az[SP_MAXCH][4], vz[SP_MAXCH][4], al[SP_MAXCH][4], vl[SP_MAXCH][4];
#pragma unroll
    for (int q = 0; q < SP_MAXCH; ++q) {
        const int c4 = lane + 64 * q;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), a0 = p0, v0 = p0;
        if (c4 < nc4) {
            const size_t o = (size_t)item * W + 4 * c4;
            p0 = *reinterpret_cast<const float4*>(P + o);
            a0 = *reinterpret_cast<const float4*>(A + o);
            if (mom) v0 = *reinterpret_cast<const float4*>(V + o);
        }
        pz[q][0] = p0.x; pz[q][1] = p0.y; pz[q][2] = p0.z; pz[q][3] = p0.w;
        az[q][0] = a0.x; az[q][1] = a0.y; az[q][2] = a0.z; az[q][3] = a0.w;
        vz[q][0] = v0.x; vz[q][1] = v0.y; vz[q][2] = v0.z; vz[q][3] = v0.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pc[q][e] = pz[q][e]; al[q][e] = az[q][e]; vl[q][e] = vz[q][e]; }
    }
    // all 64 lanes take part in every ballot; the per-match work below is predicated per lane
    for (int base = lo & ~63; base <= k; base += 64) {
        const int j = base + lane;
        unsigned long long mask = __ballot(j >= lo && j <= k && occ[j] == item);
        while (mask) {   // ascending occurrence order
            const int bit = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const int jj = base + bit;
            const float* grow = (jj < B) ? gx + (size_t)jj * W : gy + (size_t)(jj - B) * W;
#pragma unroll
            for (int q = 0; q < SP_MAXCH; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < nc4) {
                    const float4 g4 = *reinterpret_cast<const float4*>(grow + 4 * c4);
                    const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float an = az[q][e] + gv[e] * gv[e];
                        const float gs = gv[e] / sqrtf(an + G4R_EPS_ADAGRAD);
                        const float delta = (m.lmbd > 0.f) ? m.lr * (gs + m.lmbd * pz[q][e]) : m.lr * gs;
                        al[q][e] = an;
                        if (mom) {
                            const float v2 = m.mom * vz[q][e] - delta;
                            vl[q][e] = v2;
                            pc[q][e] = pc[q][e] + v2;
                        } else {
                            pc[q][e] = pc[q][e] - delta;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SP_MAXCH; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < nc4) {
            const size_t o = (size_t)item * W + 4 * c4;
            *reinterpret_cast<float4*>(P + o) = make_float4(pc[q][0], pc[q][1], pc[q][2], pc[q][3]);
            *reinterpret_cast<float4*>(A + o) = make_float4(al[q][0], al[q][1], al[q][2], al[q][3]);
            if (mom) *reinterpret_cast<float4*>(V + o) = make_float4(vl[q][0], vl[q][1], vl[q][2], vl[q][3]);
        }
    }
}

__global__ __launch_bounds__(256) void k_sparse_update(const DevModel* __restrict__ mp, int nblk_occ) {
    const DevModel& m = *mp;
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
    const int k = blockIdx.x * 4 + wid;
    if (k >= R) return;
    const int item = m.occ_idx[k];
    if (item < 0) return;
    const bool constrained = (m.embed_mode == G4R_EMBED_CONSTRAINED);
    // occurrence range sharing a table with k: constrained -> all of X|Y|samples ; separate -> X alone, Y|samples alone
    const int lo = (constrained || k < B) ? 0 : B;
    const int hi = (constrained || k >= B) ? R : B;
    // ---- is there a later occurrence of the same item?  then that wave owns the row
    for (int base = (k + 1) & ~63; base < hi; base += 64) {
        const int j = base + lane;
        if (__ballot(j > k && j < hi && m.occ_idx[j] == item)) return;
    }
    if (k < B && !constrained) {
        sparse_row_update(m, m.E, m.accE, m.velE, item, m.Ein, m.occ_idx, lo, k, m.dSx, nullptr, B, lane);
        return;
    }
    sparse_row_update(m, m.Wy, m.accWy, m.velWy, item, m.Dtop, m.occ_idx, lo, k, m.dSx, m.dSy, B, lane);
    if (k >= B) {
        // ---- output bias By: occurrences among Y|samples only (gru4rec.py:486-489)
        float p = m.By[item];
        const float pz = p, az = m.accBy[item], vz = (m.mom > 0.f) ? m.velBy[item] : 0.f;
        float al = az, vl = vz;
        for (int base = B & ~63; base <= k; base += 64) {
            const int j = base + lane;
            unsigned long long mask = __ballot(j >= B && j <= k && m.occ_idx[j] == item);
            while (mask) {
                const int bit = __ffsll((unsigned long long)mask) - 1;
                mask &= mask - 1;
                const float g = m.dSBy[base + bit - B];
                const float an = az + g * g;
                const float gs = g / sqrtf(an + G4R_EPS_ADAGRAD);
                const float delta = (m.lmbd > 0.f) ? m.lr * (gs + m.lmbd * pz) : m.lr * gs;
                al = an;
                if (m.mom > 0.f) { const float v2 = m.mom * vz - delta; vl = v2; p = p + v2; }
                else p = p - delta;
            }
        }
        if (lane == 0) {
            m.By[item] = p;
            m.accBy[item] = al;
            if (m.mom > 0.f) m.velBy[item] = vl;
        }
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
