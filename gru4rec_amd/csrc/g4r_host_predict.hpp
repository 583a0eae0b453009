// g4r_host_predict.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: prediction and evaluation: g4r_predict_*, g4r_rank_targets, g4r_evaluate.
// ------------------------------------------------------------------------------------------------ prediction
int g4r_predict_begin(g4r_model* m, int32_t batch) {
    if (!m || batch < 1) return fail("bad batch");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipStreamSynchronize(m->stream));
    DevModel& d = m->dm;
    if (batch != m->pbatch) {
        for (int l = 0; l < d.n_layers; ++l) {
            dfree(m, m->pH[l][0]); dfree(m, m->pH[l][1]); dfree(m, m->phout[l]);
            dfree(m, m->pVc[l]); dfree(m, m->pz[l]); dfree(m, m->pHr[l]);
            if (dalloc(m, &m->pVc[l], (size_t)batch * d.D[l]) || dalloc(m, &m->pz[l], (size_t)batch * d.D[l]) ||
                dalloc(m, &m->pHr[l], (size_t)batch * d.D[l]))
                return -1;
            if (dalloc(m, &m->pH[l][0], (size_t)batch * d.D[l]) || dalloc(m, &m->pH[l][1], (size_t)batch * d.D[l]) ||
                dalloc(m, &m->phout[l], (size_t)batch * d.D[l]))
                return -1;
        }
        dfree(m, m->p_in); dfree(m, m->p_tgt); dfree(m, m->p_keep); dfree(m, m->p_zero); dfree(m, m->p_ranks); dfree(m, m->p_cnt);
        if (dalloc(m, &m->p_in, batch) || dalloc(m, &m->p_tgt, batch) || dalloc(m, &m->p_keep, batch) ||
            dalloc(m, &m->p_zero, batch) || dalloc(m, &m->p_ranks, batch) || dalloc(m, &m->p_cnt, 2 * (size_t)batch))
            return -1;
        m->pbatch = batch;
    } else {
        for (int l = 0; l < d.n_layers; ++l)
            for (int q = 0; q < 2; ++q) HIPCHK(hipMemsetAsync(m->pH[l][q], 0, (size_t)batch * d.D[l] * sizeof(float), m->stream));
    }
    m->ppar = 0;
    m->tie_ctr = 0;
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

int g4r_predict_hidden(g4r_model* m, const uint8_t* zero_mask, int32_t n_mask, const int32_t* keep_rows, int32_t n_keep) {
    if (!m || !m->pbatch) return fail("g4r_predict_begin first");
    HIPCHK(hipSetDevice(m->cfg.device));
    DevModel& d = m->dm;
    const int PB = m->pbatch;
    if (zero_mask) {
        if (n_mask < 0 || n_mask > PB) return fail("zero_mask is longer than the prediction batch (g4r_predict_begin)");
        std::vector<unsigned char> zm(PB, 0);      // rows past the mask keep their state
        memcpy(zm.data(), zero_mask, (size_t)n_mask);
        HIPCHK(hipMemcpyAsync(m->p_zero, zm.data(), PB, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        for (int l = 0; l < d.n_layers; ++l)
            hipLaunchKernelGGL(k_zero_rows, dim3(cdiv((long long)PB * d.D[l], 256)), dim3(256), 0, m->stream, m->pH[l][m->ppar],
                               (const unsigned char*)m->p_zero, PB, d.D[l]);
    }
    if (keep_rows) {
        if (n_keep < 0 || n_keep > PB) return fail("n_keep out of range");
        std::vector<int> mp(PB, -1);
        for (int j = 0; j < n_keep; ++j) mp[j] = keep_rows[j];
        HIPCHK(hipMemcpyAsync(m->p_keep, mp.data(), PB * sizeof(int), hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        for (int l = 0; l < d.n_layers; ++l)
            hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((long long)PB * d.D[l], 256)), dim3(256), 0, m->stream, m->pH[l][m->ppar ^ 1],
                               (const float*)m->pH[l][m->ppar], (const int*)m->p_keep, PB, d.D[l]);
        m->ppar ^= 1;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

struct StreamRank;
static int predict_forward(g4r_model* m, const int* d_in_idx, int mrows, const int* d_items, int64_t n_sel, const StreamRank* stream);

int g4r_predict_step(g4r_model* m, const int32_t* in_idx, int32_t mrows, const int32_t* item_idx, int64_t n_sel,
                     float* out_scores) {
    if (!m || !in_idx) return fail("null argument");
    if (!m->pbatch) return fail("g4r_predict_begin first");
    if (mrows < 1 || mrows > m->pbatch) return fail("mrows out of range");
    HIPCHK(hipSetDevice(m->cfg.device));
    DevModel& d = m->dm;
    if (!item_idx) n_sel = d.n_items;
    if (n_sel < 1) return fail("n_sel must be positive");
    for (int i = 0; i < mrows; ++i)
        if (in_idx[i] < 0 || in_idx[i] >= d.n_items) return fail("input item index out of range");
    HIPCHK(hipMemcpyAsync(m->p_in, in_idx, mrows * sizeof(int), hipMemcpyHostToDevice, m->stream));
    if (item_idx) {
        if (n_sel > m->p_items_cap) {
            dfree(m, m->p_items);
            if (dalloc(m, &m->p_items, (size_t)n_sel, false)) return -1;
            m->p_items_cap = n_sel;
        }
        for (int64_t i = 0; i < n_sel; ++i)
            if (item_idx[i] < 0 || item_idx[i] >= d.n_items) return fail("item index out of range");
        HIPCHK(hipMemcpyAsync(m->p_items, item_idx, n_sel * sizeof(int), hipMemcpyHostToDevice, m->stream));
    }
    if (predict_forward(m, m->p_in, mrows, item_idx ? (const int*)m->p_items : (const int*)nullptr, n_sel, nullptr)) return -1;
    const int64_t ldo = m->p_ldo;
    if (out_scores) {
        HIPCHK(hipMemcpy2DAsync(out_scores, n_sel * sizeof(float), m->p_scores, ldo * sizeof(float), n_sel * sizeof(float), mrows,
                                hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

int g4r_rank_targets(g4r_model* m, const int32_t* target_col, int32_t mrows, int64_t col_begin, int32_t mode, float* ranks) {
    if (!m || !target_col || !ranks) return fail("null argument");
    if (!m->p_scores || mrows < 1 || mrows > m->pbatch) return fail("no scores / mrows out of range");
    if (mode < 0 || mode > G4R_RANK_TIEBREAKING) return fail("unknown rank mode");
    for (int i = 0; i < mrows; ++i)
        if (target_col[i] < 0 || target_col[i] >= m->p_nsel) return fail("target column out of range");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipMemcpyAsync(m->p_tgt, target_col, mrows * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_rank_rows, dim3(mrows), dim3(256), 0, m->stream, (const float*)m->p_scores, (long long)m->p_nsel,
                       (long long)m->p_ldo, (const int*)m->p_tgt, (long long)col_begin, (int)mode, m->p_ranks,
                       (unsigned long long)m->cfg.seed, m->tie_ctr++);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ranks, m->p_ranks, mrows * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

// forward GRU + scores of `mrows` rows whose input items sit on the device (shared by g4r_predict_step / g4r_evaluate)
// stream = nullptr: scores of all candidates go to p_scores (final activation applied).  Otherwise (evaluation with an
// element-wise final activation) nothing is materialised: stream->tgt lists the target item of every row; their scores are
// computed first (mrows x mrows tile, diagonal used), then every candidate tile is compared with them on the fly and
// p_ranks receives the ranks (stream->mode, candidates from column stream->col_begin on).
struct StreamRank { const int* tgt; long long col_begin; int mode; const int* tie_col; unsigned tie_ctr; };
static int predict_forward(g4r_model* m, const int* d_in_idx, int mrows, const int* d_items, int64_t n_sel, const StreamRank* stream) {
    DevModel& d = m->dm;
    const int64_t ldo = stream ? ((mrows + 3) & ~3) : ((n_sel + 3) & ~3LL);
    const int64_t need = stream ? (int64_t)m->pbatch * ((m->pbatch + 3) & ~3) : (int64_t)m->pbatch * ldo;
    if (need > m->p_scores_cap) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(m, m->p_scores);
        if (dalloc(m, &m->p_scores, (size_t)need, false)) return -1;
        m->p_scores_cap = need;
    }
    for (int l = 0; l < d.n_layers; ++l) {
        GruFwdPredict pa;
        pa.in_idx = (GP(const int))d_in_idx;
        pa.ysrc = (GP(const float))(l > 0 ? m->phout[l - 1] : nullptr);
        pa.Hcur = (GP(const float))m->pH[l][m->ppar];
        pa.Hnext = (GP(float))m->pH[l][m->ppar ^ 1];
        pa.hout = (GP(float))m->phout[l];
        pa.Vc = (GP(float))m->pVc[l]; pa.z = (GP(float))m->pz[l]; pa.Hr = (GP(float))m->pHr[l];
        pa.M = mrows;
        if (wide_layer(d.D[l]))
            hipLaunchKernelGGL(k_gru_p1_n64, dim3(cdiv(3 * d.D[l], 64), cdiv(mrows, GT_BM)), dim3(GT_NTH_FEW), SMEM_P1_N64, m->stream,
                               (const DevModel*)m->d_dm, (StepState*)nullptr, l, 0, 0, pa);
        else
            hipLaunchKernelGGL(k_gru_p1_n32, dim3(cdiv(3 * d.D[l], GT_BN), cdiv(mrows, GT_BM)), dim3(GT_NTH_FEW), SMEM_P1, m->stream,
                               (const DevModel*)m->d_dm, (StepState*)nullptr, l, 0, 0, pa);
        {
            const dim3 g2(cdiv(d.D[l], GT_BN), cdiv(mrows, GT_BM));
            // (geometry from the TRAINING batch, not from this call's rows: the fp32 summation order of the hidden state must not depend
            // on the evaluation batch size, nor differ between training and prediction)
            if (deep_geometry(m->p2_geo_env, m->n_cu, d.D[l], d.B))
                hipLaunchKernelGGL(k_gru_p2_w8d, g2, dim3(512), SMEM_P2_256, m->stream, (const DevModel*)m->d_dm, (StepState*)nullptr, l, 0, pa);
            else hipLaunchKernelGGL(k_gru_p2_w4, g2, dim3(GT_NTH), SMEM_NN, m->stream, (const DevModel*)m->d_dm, (StepState*)nullptr, l, 0, pa);
        }
    }
    m->ppar ^= 1;
    const bool sm = (d.final_act == G4R_ACT_SOFTMAX || d.final_act == G4R_ACT_SOFTMAX_LOGIT);   // gru4rec.py:499-500
    const float* hsrc = (const float*)m->phout[d.n_layers - 1];
    if (stream) {
        if (sm) return fail("internal: streaming ranks need an element-wise final activation");
        hipLaunchKernelGGL(k_score_store, dim3(cdiv(mrows, 32), cdiv(mrows, SC_BM)), dim3(256), m->smem_score, m->stream, (const DevModel*)m->d_dm,
                           hsrc, (int)mrows, stream->tgt, (long long)mrows, m->p_scores, (long long)ldo, 1, (int*)nullptr, 0LL, (const int*)nullptr, 0u);
        hipLaunchKernelGGL(k_score_count, dim3(cdiv(n_sel, 32), cdiv(mrows, SC_BM)), dim3(256), m->smem_score, m->stream, (const DevModel*)m->d_dm,
                           hsrc, (int)mrows, d_items, (long long)n_sel, m->p_scores, (long long)ldo, 1, m->p_cnt, stream->col_begin,
                           stream->mode == G4R_RANK_TIEBREAKING ? stream->tie_col : (const int*)nullptr, stream->tie_ctr);
        hipLaunchKernelGGL(k_rank_counts, dim3(cdiv(mrows, 256)), dim3(256), 0, m->stream, m->p_cnt, (int)mrows, stream->mode, m->p_ranks);
        HIPCHK(hipGetLastError());
        m->p_nsel = 0; m->p_ldo = ldo;        // no score matrix to read back
        return 0;
    }
    hipLaunchKernelGGL(k_score_store, dim3(cdiv(n_sel, 32), cdiv(mrows, SC_BM)), dim3(256), m->smem_score, m->stream, (const DevModel*)m->d_dm,
                       hsrc, (int)mrows, d_items, (long long)n_sel, m->p_scores, (long long)ldo, sm ? 0 : 1, (int*)nullptr, 0LL, (const int*)nullptr, 0u);
    if (sm) hipLaunchKernelGGL(k_softmax_rows, dim3(mrows), dim3(256), 0, m->stream, m->p_scores, (long long)n_sel, (long long)ldo);
    HIPCHK(hipGetLastError());
    m->p_nsel = n_sel; m->p_ldo = ldo;
    return 0;
}

int g4r_evaluate(g4r_model* m, const int32_t* in_idx, const int32_t* out_idx, const uint8_t* reset, const int32_t* M, int64_t T,
                 int32_t batch, const int64_t* compact_steps, const int32_t* compact_maps, int64_t n_compact,
                 const int32_t* items, int64_t n_items_sel, const int32_t* cutoffs, int32_t n_cut, int32_t mode,
                 double* recall_sum, double* mrr_sum, int64_t* n_events) {
    if (!m || !in_idx || !out_idx || !reset || !M || !cutoffs || !recall_sum || !mrr_sum || !n_events) return fail("null argument");
    if (T < 0 || batch < 1 || n_cut < 1 || n_cut > 64) return fail("bad evaluation sizes");
    if (mode < 0 || mode > G4R_RANK_TIEBREAKING) return fail("unknown rank mode");
    if (n_compact > 0 && (!compact_steps || !compact_maps)) return fail("compaction arrays missing");
    DevModel& d = m->dm;
    const int B = batch;
    for (int64_t i = 0; i < T * B; ++i)
        if (in_idx[i] < 0 || in_idx[i] >= d.n_items || out_idx[i] < 0 || out_idx[i] >= d.n_items) return fail("plan item index out of range");
    for (int64_t i = 0; i < n_items_sel; ++i)
        if (items[i] < 0 || items[i] >= d.n_items) return fail("item index out of range");
    if (g4r_predict_begin(m, batch)) return -1;            // fresh (zero) hidden state, scratch for `batch` rows
    int *e_in = nullptr, *e_out = nullptr, *e_M = nullptr, *e_maps = nullptr, *e_items = nullptr, *e_cand = nullptr, *e_cut = nullptr, *e_iota = nullptr;
    unsigned char* e_reset = nullptr;
    double* e_acc = nullptr;            // [rec(n_cut) | mrr(n_cut)]
    long long* e_n = nullptr;
    const size_t TB = (size_t)std::max<int64_t>(T, 1) * B;
    auto cleanup = [&]() {
        dfree(m, e_in); dfree(m, e_out); dfree(m, e_M); dfree(m, e_maps); dfree(m, e_items); dfree(m, e_cand); dfree(m, e_cut);
        dfree(m, e_iota); dfree(m, e_reset); dfree(m, e_acc); dfree(m, e_n);
    };
    if (dalloc(m, &e_in, TB, false) || dalloc(m, &e_out, TB, false) || dalloc(m, &e_reset, TB, false) ||
        dalloc(m, &e_maps, (size_t)std::max<int64_t>(n_compact, 1) * B, false) || dalloc(m, &e_cut, n_cut, false) ||
        dalloc(m, &e_acc, 2 * (size_t)n_cut) || dalloc(m, &e_n, 1) || dalloc(m, &e_iota, B, false) ||
        (items && (dalloc(m, &e_items, (size_t)n_items_sel, false) || dalloc(m, &e_cand, (size_t)B + n_items_sel, false)))) {
        cleanup();
        return -1;
    }
#define EVCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(std::string(#x ": ") + hipGetErrorString(e_)); } } while (0)
    hipStream_t s = m->stream;
    if (T > 0) {
        EVCHK(hipMemcpyAsync(e_in, in_idx, TB * sizeof(int), hipMemcpyHostToDevice, s));
        EVCHK(hipMemcpyAsync(e_out, out_idx, TB * sizeof(int), hipMemcpyHostToDevice, s));
        EVCHK(hipMemcpyAsync(e_reset, reset, TB, hipMemcpyHostToDevice, s));
    }
    if (n_compact > 0) EVCHK(hipMemcpyAsync(e_maps, compact_maps, (size_t)n_compact * B * sizeof(int), hipMemcpyHostToDevice, s));
    EVCHK(hipMemcpyAsync(e_cut, cutoffs, n_cut * sizeof(int), hipMemcpyHostToDevice, s));
    if (items) EVCHK(hipMemcpyAsync(e_items, items, (size_t)n_items_sel * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_iota, dim3(cdiv(B, 256)), dim3(256), 0, s, e_iota, B);
    const bool sm_act = (d.final_act == G4R_ACT_SOFTMAX || d.final_act == G4R_ACT_SOFTMAX_LOGIT);
    const bool streaming = !sm_act && !getenv("G4R_EVAL_MATERIALIZE");
    int64_t ci = 0;
    for (int64_t t = 0; t < T; ++t) {
        const int Mt = M[t];
        if (Mt < 1 || Mt > B) { cleanup(); return fail("plan M out of range"); }
        // rows of exhausted slots are dropped before this step (evaluation.py:138; gru4rec.py:647-651 for the same plan format)
        while (ci < n_compact && compact_steps[ci] == t) {
            for (int l = 0; l < d.n_layers; ++l)
                hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((long long)B * d.D[l], 256)), dim3(256), 0, s, m->pH[l][m->ppar ^ 1],
                                   (const float*)m->pH[l][m->ppar], (const int*)(e_maps + ci * B), B, d.D[l]);
            m->ppar ^= 1;
            ++ci;
        }
        const int* tgt = e_out + t * B;
        const int* cand = nullptr;
        int64_t n_sel = d.n_items;
        if (items) {
            hipLaunchKernelGGL(k_eval_candidates, dim3(cdiv((long long)Mt + n_items_sel, 256)), dim3(256), 0, s, e_cand, tgt, Mt,
                               (const int*)e_items, (long long)n_items_sel);
            cand = e_cand;
            n_sel = Mt + n_items_sel;
        }
        if (streaming) {
            // element-wise final activation: candidate tiles are ranked against the target score as they are produced
            // column of row i's target in the candidate list: i when [targets | items] are scored, the target item otherwise
            const StreamRank sr = {tgt, items ? (long long)Mt : 0LL, (int)mode, items ? (const int*)e_iota : tgt, (unsigned)t};
            if (predict_forward(m, e_in + t * B, Mt, cand, n_sel, &sr)) { cleanup(); return -1; }
        } else {
            // softmax needs the whole row first (max, sum): scores are materialised, then ranked
            if (predict_forward(m, e_in + t * B, Mt, cand, n_sel, nullptr)) { cleanup(); return -1; }
            hipLaunchKernelGGL(k_rank_rows, dim3(Mt), dim3(256), 0, s, (const float*)m->p_scores, (long long)m->p_nsel, (long long)m->p_ldo,
                               items ? (const int*)e_iota : tgt, items ? (long long)Mt : 0LL, (int)mode, m->p_ranks,
                               (unsigned long long)m->cfg.seed, (unsigned)t);
        }
        hipLaunchKernelGGL(k_eval_accum, dim3(1), dim3(256), 0, s, (const float*)m->p_ranks, Mt, (const int*)e_cut, (int)n_cut, e_acc,
                           e_acc + n_cut, e_n);
        // hidden rows of sessions that ended with this step start from zero (evaluation.py:137)
        for (int l = 0; l < d.n_layers; ++l)
            hipLaunchKernelGGL(k_zero_rows, dim3(cdiv((long long)Mt * d.D[l], 256)), dim3(256), 0, s, m->pH[l][m->ppar],
                               (const unsigned char*)(e_reset + t * B), Mt, d.D[l]);
    }
    EVCHK(hipGetLastError());
    std::vector<double> acc(2 * (size_t)n_cut);
    long long n = 0;
    EVCHK(hipMemcpyAsync(acc.data(), e_acc, acc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    EVCHK(hipMemcpyAsync(&n, e_n, sizeof(n), hipMemcpyDeviceToHost, s));
    EVCHK(hipStreamSynchronize(s));
#undef EVCHK
    for (int c = 0; c < n_cut; ++c) { recall_sum[c] = acc[c]; mrr_sum[c] = acc[n_cut + c]; }
    *n_events = n;
    cleanup();
    return 0;
}
