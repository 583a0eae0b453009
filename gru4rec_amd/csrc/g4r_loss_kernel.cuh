// g4r_loss_kernel.cuh -- part of g4r_step_kernels.cuh (included there, in order; needs its prelude).  Holds k_loss_rows: final activation, loss and d cost / d s per score row.
#pragma once
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
