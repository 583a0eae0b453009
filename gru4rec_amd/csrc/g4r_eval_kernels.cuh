// Prediction / evaluation kernels (gfx950): replaces the compiled `self.predict` of
// gru4rec.py:691-711 and the evaluate function of evaluation.py:54-76.
//   k_gru_p1/p2 (train = 0) forward GRU step without dropout / reset        (g4r_step_kernels.cuh)
//   k_score_all             scores[m, n_sel] = h Wy[items]^T + By[items]     gru4rec.py:499-505
//   k_softmax_rows          final_act == softmax over the selected items     gru4rec.py:193-195
//   k_rank_rows             rank of the target among the other scores        evaluation.py:56-65
#pragma once
#include "g4r_step_kernels.cuh"

#define SC_BM 128
#define SC_KC 128

// COUNT = false: out[row, n] = score.  COUNT = true (streaming evaluation, no score matrix): `out` holds the target score of
// every row (out[row * ldo + row], produced by a COUNT = false launch over the target items, i.e. by the very same
// k-ordered MFMA chain, so equal scores compare equal); the tile's scores are compared with it and the number of
// candidates in columns >= col_begin that are greater / equal is added to cnt[2 * row], cnt[2 * row + 1] (integer atomics).
// tie_col != nullptr (mode 'tiebreaking'): score (row, column n) is moved by tie_noise(row, n) before the comparison and the
// row's target score by tie_noise(row, tie_col[row]) -- tie_col[row] is the column the target occupies in the candidate list.
template <int TN, bool COUNT = false>
__global__ __launch_bounds__(256) void k_score_all(const DevModel* __restrict__ mp, const float* h, int mrows, const int* item_idx,
                                                   long long n_sel, float* out, long long ldo, int apply_act,
                                                   int* cnt = nullptr, long long col_begin = 0, const int* tie_col = nullptr,
                                                   unsigned tie_ctr = 0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevModel& m = *mp;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int D = m.Dtop;
    const int ldk = SC_KC + 2;
    float* sA = smem;
    float* sB = sA + SC_BM * ldk;
    int* sItem = reinterpret_cast<int*>(sB + TN * ldk);
    const long long n0 = (long long)blockIdx.x * TN;
    const int rbase = blockIdx.y * SC_BM;
    if (tid < TN) {
        const long long n = n0 + tid;
        int item = -1;
        if (n < n_sel) item = item_idx ? item_idx[n] : (int)n;
        sItem[tid] = item;
    }
    __syncthreads();
    constexpr int CT = TN / 16;
    f32x4 acc[2][CT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc0 = 0; kc0 < D; kc0 += SC_KC) {
        const int kc = min(SC_KC, D - kc0), kc4 = kc >> 2;
        for (int e = tid; e < SC_BM * kc4; e += 256) {
            const int i = e / kc4, c4 = e - i * kc4, row = rbase + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < mrows) v = ld4(h + (size_t)row * D + kc0 + 4 * c4);
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
    if constexpr (!COUNT) {
#pragma unroll
        for (int cj = 0; cj < CT; ++cj) {
            const long long n = n0 + 16 * cj + li;
            const int item = sItem[16 * cj + li];
            const float add = item >= 0 ? m.By[item] : 0.f;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int row = rbase + 32 * wid + 16 * ri + 4 * lg + rg;
                    if (row < mrows && n < n_sel) {
                        float v = acc[ri][cj][rg] + add;
                        if (apply_act) v = act_fwd(m.final_act, m.fa_p0, m.fa_p1, v);
                        out[(size_t)row * ldo + n] = v;
                    }
                }
        }
    } else {
        float add[CT];
#pragma unroll
        for (int cj = 0; cj < CT; ++cj) { const int item = sItem[16 * cj + li]; add[cj] = item >= 0 ? m.By[item] : 0.f; }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = rbase + 32 * wid + 16 * ri + 4 * lg + rg;
                float t = out[(size_t)min(row, mrows - 1) * ldo + min(row, mrows - 1)];
                if (tie_col) t += tie_noise(m.seed, tie_ctr, row, tie_col[min(row, mrows - 1)]);
                float gt = 0.f, eq = 0.f;
#pragma unroll
                for (int cj = 0; cj < CT; ++cj) {
                    const long long n = n0 + 16 * cj + li;
                    float v = acc[ri][cj][rg] + add[cj];
                    if (apply_act) v = act_fwd(m.final_act, m.fa_p0, m.fa_p1, v);
                    if (tie_col) v += tie_noise(m.seed, tie_ctr, row, n);
                    const bool in = n < n_sel && n >= col_begin;
                    gt += (in && v > t) ? 1.f : 0.f;
                    eq += (in && v == t) ? 1.f : 0.f;
                }
                // the 16 lanes that share this row (same lg): quad swaps, mirrored half row, mirrored row
                gt += dpp_mov<0xB1>(gt); gt += dpp_mov<0x4E>(gt); gt += dpp_mov<0x141>(gt); gt += dpp_mov<0x140>(gt);
                eq += dpp_mov<0xB1>(eq); eq += dpp_mov<0x4E>(eq); eq += dpp_mov<0x141>(eq); eq += dpp_mov<0x140>(eq);
                if (li == 0 && row < mrows) {
                    if (gt != 0.f) atomicAdd(cnt + 2 * row, (int)gt);
                    if (eq != 0.f) atomicAdd(cnt + 2 * row + 1, (int)eq);
                }
            }
    }
}

// ranks from the streamed counts (evaluation.py:62-65); clears the counters for the next step
__global__ __launch_bounds__(256) void k_rank_counts(int* cnt, int mrows, int mode, float* ranks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= mrows) return;
    const float gt = (float)cnt[2 * i], eq = (float)cnt[2 * i + 1];
    cnt[2 * i] = 0; cnt[2 * i + 1] = 0;
    float r;
    if (mode == G4R_RANK_CONSERVATIVE) r = gt + eq;
    else if (mode == G4R_RANK_MEDIAN) r = gt + 0.5f * (eq - 1.f) + 1.f;
    else r = gt + 1.f;      // STANDARD, TIEBREAKING
    ranks[i] = r;
}

// in-place softmax over n_sel columns of each row (one 256-thread workgroup per row)
__global__ __launch_bounds__(256) void k_softmax_rows(float* sc, long long n_sel, long long ldo) {
    __shared__ float red[8];
    float* row = sc + (size_t)blockIdx.x * ldo;
    float mx = -INFINITY;
    for (long long j = threadIdx.x; j < n_sel; j += 256) mx = fmaxf(mx, row[j]);
    mx = block_max_256(mx, red);
    float sm = 0.f;
    for (long long j = threadIdx.x; j < n_sel; j += 256) sm += expf(row[j] - mx);
    sm = block_sum_256(sm, red);
    for (long long j = threadIdx.x; j < n_sel; j += 256) row[j] = expf(row[j] - mx) / sm;
}

// ranks (evaluation.py:62-65): others = columns [col_begin, n_sel); target = column target_col[row]
__global__ __launch_bounds__(256) void k_rank_rows(const float* sc, long long n_sel, long long ldo, const int* target_col,
                                                   long long col_begin, int mode, float* ranks, unsigned long long seed, unsigned tie_ctr) {
    __shared__ float red[8];
    const float* row = sc + (size_t)blockIdx.x * ldo;
    const bool tie = mode == G4R_RANK_TIEBREAKING;      // evaluation.py:55: yhat += uniform * 1e-10 (fp32), then as STANDARD
    float t = row[target_col[blockIdx.x]];
    if (tie) t += tie_noise(seed, tie_ctr, blockIdx.x, target_col[blockIdx.x]);
    float gt = 0.f, eq = 0.f;
    for (long long j = col_begin + threadIdx.x; j < n_sel; j += 256) {
        float v = row[j];
        if (tie) v += tie_noise(seed, tie_ctr, blockIdx.x, j);
        gt += (v > t) ? 1.f : 0.f;
        eq += (v == t) ? 1.f : 0.f;
    }
    gt = block_sum_256(gt, red);
    eq = block_sum_256(eq, red);
    if (threadIdx.x == 0) {
        float r;
        if (mode == G4R_RANK_CONSERVATIVE) r = gt + eq;
        else if (mode == G4R_RANK_MEDIAN) r = gt + 0.5f * (eq - 1.f) + 1.f;
        else r = gt + 1.f;
        ranks[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void k_zero_rows(float* H, const unsigned char* zero_mask, int nrows, int W) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * W) return;
    if (zero_mask[e / W]) H[e] = 0.f;
}

// ---- device-resident evaluation (evaluation.py:15-147 as one call, g4r_evaluate) -------------------------------
// candidate list of one step when `items` are given: [targets of the M rows | items]   (evaluation.py:103-104)
__global__ __launch_bounds__(256) void k_eval_candidates(int* cand, const int* tgt, int M, const int* items, long long n_items_sel) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < M) cand[e] = tgt[e];
    else if (e < M + n_items_sel) cand[e] = items[e - M];
}
__global__ __launch_bounds__(256) void k_iota(int* p, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) p[e] = e;
}
// recall[c] += #(rank <= cut_c), mrr[c] += sum over those rows of 1 / rank, n += M  (evaluation.py:66-72,106-114);
// one workgroup, fixed summation order, double accumulators
__global__ __launch_bounds__(256) void k_eval_accum(const float* ranks, int M, const int* cuts, int n_cut, double* rec, double* mrr,
                                                    long long* n) {
    __shared__ double sh[2][256];
    for (int c = 0; c < n_cut; ++c) {
        const float cut = (float)cuts[c];
        double h = 0.0, r = 0.0;
        for (int i = threadIdx.x; i < M; i += 256) {
            const float rk = ranks[i];
            if (rk <= cut) { h += 1.0; r += 1.0 / (double)rk; }
        }
        sh[0][threadIdx.x] = h; sh[1][threadIdx.x] = r;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { rec[c] += sh[0][0]; mrr[c] += sh[1][0]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n += M;
}

// 16x16x4 fp32 MFMA operand/accumulator layout self-test: C = A(16xK) * B(Kx16), asymmetric inputs
__global__ void k_selftest_mfma(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = mfma16(A[li * K + k + lg], Bm[(k + lg) * 16 + li], acc);
    for (int rg = 0; rg < 4; ++rg) C[(4 * lg + rg) * 16 + li] = acc[rg];
}

// 32x32x2 fp32 MFMA operand/accumulator layout self-test: C(32x32) = A(32xK) * B(Kx32)
__global__ void k_selftest_mfma32(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x & 63, l32 = lane & 31, lh = lane >> 5;
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int k = 0; k < K; k += 2) acc = mfma32(A[l32 * K + k + lh], Bm[(k + lh) * 32 + l32], acc);
    for (int r = 0; r < 16; ++r) C[(8 * (r >> 2) + 4 * lh + (r & 3)) * 32 + l32] = acc[r];
}

// explicit instantiations: the launching host code is not visible to the device pass
template __global__ void k_sparse_update<1, false>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update<1, true>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update<2, false>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update<2, true>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update_generic<1>(const DevModel*, StepState*, int, int);
template __global__ void k_sparse_update_generic<2>(const DevModel*, StepState*, int, int);
template __global__ void k_loss_rows<false, 0, 1>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 0, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 1, 1>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 1, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 2, 1>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 2, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 3, 1>(const DevModel*, StepState*);
template __global__ void k_loss_rows<false, 3, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<true, 0, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<true, 1, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<true, 2, 4>(const DevModel*, StepState*);
template __global__ void k_loss_rows<true, 3, 4>(const DevModel*, StepState*);
template __global__ void k_sparse_update<4, false>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update<4, true>(const DevModel*, StepState*, int);
template __global__ void k_sparse_update_generic<4>(const DevModel*, StepState*, int, int);
template __global__ void k_update<1, 32, false>(const DevModel*, StepState*, const DenseTile*, int, int);
template __global__ void k_update<1, 32, true>(const DevModel*, StepState*, const DenseTile*, int, int);
template __global__ void k_update<2, 32, false>(const DevModel*, StepState*, const DenseTile*, int, int);
template __global__ void k_update<2, 32, true>(const DevModel*, StepState*, const DenseTile*, int, int);
template __global__ void k_dense_grad<32>(const DevModel*, StepState*, const DenseTile*);
template __global__ void k_score_fwd<GT_BN, GT_BK>(const DevModel*, StepState*);
template __global__ void k_score_fwd<32, 64>(const DevModel*, StepState*);
template __global__ void k_score_fwd<64, 32, T2_BK>(const DevModel*, StepState*);
template __global__ void k_score_fwd<64, 32, 3>(const DevModel*, StepState*);
template __global__ void k_score_bwd<32, GT_BK>(const DevModel*, StepState*, int, int, int, int);
template __global__ void k_score_bwd<64, 64>(const DevModel*, StepState*, int, int, int, int);
template __global__ void k_gru_p1<GT_BN, P1_BK>(const DevModel*, StepState*, int, int, int, GruFwdPredict);
template __global__ void k_gru_p1<64, 256>(const DevModel*, StepState*, int, int, int, GruFwdPredict);
template __global__ void k_gru_p2<GT_NTH, GT_BK>(const DevModel*, StepState*, int, int, GruFwdPredict);
template __global__ void k_gru_p2<512, 256>(const DevModel*, StepState*, int, int, GruFwdPredict);
template __global__ void k_gru_bwd_a<GT_NTH, GT_BK>(const DevModel*, StepState*, int, int);
template __global__ void k_gru_bwd_a<512, 256>(const DevModel*, StepState*, int, int);
template __global__ void k_score_all<32, false>(const DevModel*, const float*, int, const int*, long long, float*, long long, int, int*, long long, const int*, unsigned);
template __global__ void k_score_all<32, true>(const DevModel*, const float*, int, const int*, long long, float*, long long, int, int*, long long, const int*, unsigned);
