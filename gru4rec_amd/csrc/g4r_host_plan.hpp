// g4r_host_plan.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: parameters in and out, popularity tables and the sample store, g4r_build_plan / g4r_set_plan.
// ------------------------------------------------------------------------------------------------ parameters
static int locate(g4r_model* m, const char* name, int layer, float** p, int64_t* n) {
    DevModel& d = m->dm;
    std::string s(name);
    float *base_p = d.dense_p;
    bool want_acc = false, want_vel = false, want_acc2 = false, want_cnt = false;
    if (s.rfind("acc2_", 0) == 0) { want_acc2 = true; s = s.substr(5); }
    else if (s.rfind("cnt_", 0) == 0) { want_cnt = true; s = s.substr(4); }
    else if (s.rfind("acc_", 0) == 0) { want_acc = true; s = s.substr(4); }
    else if (s.rfind("vel_", 0) == 0) { want_vel = true; s = s.substr(4); }
    if (want_vel && m->cfg.momentum <= 0.f && (s == "Wy" || s == "By" || s == "E" || (d.embed_mode == G4R_EMBED_ONEHOT && s == "Wx" && layer == 0)))
        return fail("no velocity state without momentum");
    const int64_t I = d.n_items;
    if ((want_acc2 && !d.dense_acc2) || (want_cnt && !d.dense_cnt)) return fail("this optimizer keeps no such statistic");
    if (s == "Wy") { *p = want_acc2 ? d.acc2Wy : want_cnt ? d.cntWy : want_acc ? d.accWy : (want_vel ? d.velWy : d.Wy); *n = I * d.Dtop; return 0; }
    if (s == "By") { *p = want_acc2 ? d.acc2By : want_cnt ? d.cntBy : want_acc ? d.accBy : (want_vel ? d.velBy : d.By); *n = I; return 0; }
    if (d.embed_mode == G4R_EMBED_ONEHOT && s == "Wx" && layer == 0) s = "E";    // Wx[0] is the (I, 3D) row table
    if (s == "E") {
        if (!d.E) return fail("model has no separate embedding");
        *p = want_acc2 ? d.acc2E : want_cnt ? d.cntE : want_acc ? d.accE : (want_vel ? d.velE : d.E); *n = I * d.Ein; return 0;
    }
    if (layer < 0 || layer >= d.n_layers) return fail("layer out of range");
    if (want_acc) base_p = d.dense_acc; else if (want_vel) base_p = d.dense_vel;
    else if (want_acc2) base_p = d.dense_acc2; else if (want_cnt) base_p = d.dense_cnt;
    const int D = d.D[layer], IN = d.IN[layer];
    if (s == "Wx") { *p = base_p + d.offWx[layer]; *n = (int64_t)IN * 3 * D; return 0; }
    if (s == "Wh") { *p = base_p + d.offWh[layer]; *n = (int64_t)D * D; return 0; }
    if (s == "Wrz") { *p = base_p + d.offWrz[layer]; *n = (int64_t)D * 2 * D; return 0; }
    if (s == "Bh") { *p = base_p + d.offBh[layer]; *n = 3 * D; return 0; }
    if (s == "H" && !want_acc && !want_vel) { *p = d.H[layer][m->gstep & 1]; *n = (int64_t)d.B * D; return 0; }
    return fail(std::string("unknown parameter ") + name);
}

int g4r_set_param(g4r_model* m, const char* name, int32_t layer, const float* host, int64_t count) {
    if (!m || !name || !host) return fail("null argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    float* p; int64_t n;
    if (locate(m, name, layer, &p, &n)) return -1;
    if (n != count) return fail(std::string("size mismatch for ") + name);
    HIPCHK(hipMemcpyAsync(p, host, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
    for (int g = 0; g < 2; ++g)      // a table set from the host is the new common base of its rows
        for (auto& pl : m->planes[g])
            if (pl.cur == p) HIPCHK(hipMemcpyAsync(pl.base, host, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
int g4r_get_param(g4r_model* m, const char* name, int32_t layer, float* host, int64_t count) {
    if (!m || !name || !host) return fail("null argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    float* p; int64_t n;
    if (locate(m, name, layer, &p, &n)) return -1;
    if (n != count) return fail(std::string("size mismatch for ") + name);
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(host, p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------ sampling
static int refill_store(g4r_model* m) {
    const long long n = (long long)m->gl * m->dm.ns;
    const int blocks = cdiv(cdiv(n, 4), 256);
    hipLaunchKernelGGL(k_sample_refill, dim3(blocks), dim3(256), 0, m->stream, m->d_ST, n, m->d_P, m->dm.n_items,
                       (unsigned long long)m->cfg.seed, m->refills);
    m->refills++;
    HIPCHK(hipGetLastError());
    return 0;
}

int g4r_set_popularity(g4r_model* m, const float* cum_p, const float* lq_tgt, const float* lq_smp, int64_t n) {
    if (!m || !cum_p) return fail("null argument");
    if (n != m->dm.n_items) return fail("popularity table size != n_items");
    HIPCHK(hipSetDevice(m->cfg.device));
    if (!m->d_P) { if (dalloc(m, &m->d_P, n)) return -1; }
    HIPCHK(hipMemcpyAsync(m->d_P, cum_p, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
    if (m->dm.logq != 0.f) {
        if (!lq_tgt || !lq_smp) return fail("logq > 0 needs the logQ tables");
        if (!m->d_lqt) { if (dalloc(m, &m->d_lqt, n)) return -1; if (dalloc(m, &m->d_lqs, n)) return -1; }
        HIPCHK(hipMemcpyAsync(m->d_lqt, lq_tgt, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(m->d_lqs, lq_smp, n * sizeof(float), hipMemcpyHostToDevice, m->stream));
        m->dm.lq_tgt = m->d_lqt; m->dm.lq_smp = m->d_lqs;
    }
    m->have_pop = true;
    if (sync_dm(m)) return -1;
    if (m->dm.ns > 0 && !m->store_frozen) {
        m->refills = 0;
        if (refill_store(m)) return -1;     // gru4rec.py:564 generate_samples()
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
int64_t g4r_sample_store_rows(g4r_model* m) { return m ? m->gl : -1; }
int g4r_set_sample_store(g4r_model* m, const int32_t* store, int64_t rows) {
    if (!m || !store) return fail("null argument");
    if (rows != m->gl || m->dm.ns == 0) return fail("sample store shape mismatch");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipMemcpyAsync(m->d_ST, store, (size_t)rows * m->dm.ns * sizeof(int), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->store_frozen = true;
    return 0;
}
int g4r_get_sample_store(g4r_model* m, int32_t* store, int64_t rows) {
    if (!m || !store) return fail("null argument");
    if (rows != m->gl || m->dm.ns == 0) return fail("sample store shape mismatch");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(store, m->d_ST, (size_t)rows * m->dm.ns * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------ plan
int64_t g4r_build_plan(const int32_t* off, int64_t n_sessions, const int64_t* order, const int32_t* items,
                       int32_t B, int32_t n_sample, int32_t* in_idx, int32_t* out_idx, uint8_t* reset, int32_t* M,
                       int64_t* compact_steps, int32_t* compact_maps, int64_t max_steps, int64_t max_compact,
                       int64_t* n_compact) {
    if (!off || !order || !items || B < 1) { fail("null argument"); return -1; }
    if (n_sessions < B) { fail("fewer sessions than batch_size (the reference raises IndexError here, gru4rec.py:596)"); return -1; }
    const bool write = in_idx && out_idx && reset && M;
    std::vector<int64_t> slot(B), first(B), last(B);
    for (int j = 0; j < B; ++j) { slot[j] = j; first[j] = off[order[j]]; last[j] = off[order[j] + 1]; }
    int64_t next_free = B - 1, T = 0, nc = 0;
    int cur = B;
    std::vector<char> done(B), valid(B);
    for (;;) {
        int64_t run = last[0] - first[0];
        for (int j = 1; j < cur; ++j) run = std::min(run, last[j] - first[j]);
        for (int64_t i = 0; i + 1 < run; ++i) {
            if (write) {
                if (T >= max_steps) { fail("plan buffer too small"); return -1; }
                int32_t* pi = in_idx + T * B; int32_t* po = out_idx + T * B; uint8_t* pr = reset + T * B;
                for (int j = 0; j < cur; ++j) {
                    const int64_t e = first[j] + i;
                    pi[j] = items[e]; po[j] = items[e + 1]; pr[j] = (e + 1 == last[j] - 1) ? 1 : 0;
                }
                for (int j = cur; j < B; ++j) { pi[j] = 0; po[j] = 0; pr[j] = 0; }
                M[T] = cur;
            }
            ++T;
        }
        int n_done = 0, n_valid = 0;
        for (int j = 0; j < cur; ++j) { first[j] += run - 1; done[j] = (last[j] - first[j] <= 1); }
        for (int j = 0; j < cur; ++j) if (done[j]) { slot[j] = next_free + 1 + n_done; ++n_done; }
        next_free += n_done;
        for (int j = 0; j < cur; ++j) { valid[j] = slot[j] < n_sessions; n_valid += valid[j]; }
        if (n_valid == 0 || (n_valid < 2 && n_sample == 0)) break;
        for (int j = 0; j < cur; ++j)
            if (done[j] && valid[j]) { const int64_t s = order[slot[j]]; first[j] = off[s]; last[j] = off[s + 1]; }
        if (n_valid < cur) {
            if (compact_steps && compact_maps) {
                if (nc >= max_compact) { fail("compaction buffer too small"); return -1; }
                compact_steps[nc] = T;
                int32_t* mp = compact_maps + nc * B;
                int q = 0;
                for (int j = 0; j < cur; ++j) if (valid[j]) mp[q++] = j;
                for (; q < B; ++q) mp[q] = -1;
            }
            ++nc;
            int q = 0;
            for (int j = 0; j < cur; ++j)
                if (valid[j]) { slot[q] = slot[j]; first[q] = first[j]; last[q] = last[j]; ++q; }
            cur = n_valid;
        }
    }
    if (n_compact) *n_compact = nc;
    return T;
}

static int ensure_graph(g4r_model* m);
static int ensure_head_graph(g4r_model* m);
static int sync_dense_enqueue(g4r_model* m);
static int ensure_step_graph(g4r_model* m, bool* whole);

int g4r_set_plan(g4r_model* m, const int32_t* in_idx, const int32_t* out_idx, const uint8_t* reset, const int32_t* M,
                 int64_t T, const int64_t* compact_steps, const int32_t* compact_maps, int64_t n_compact) {
    if (!m || !in_idx || !out_idx || !reset || !M || T < 1) return fail("null / empty plan");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipStreamSynchronize(m->stream));
    const int B = m->dm.B;
    dfree(m, m->d_in); dfree(m, m->d_out); dfree(m, m->d_reset); dfree(m, m->d_M); dfree(m, m->d_cmaps);
    m->d_in = m->d_out = m->d_M = m->d_cmaps = nullptr; m->d_reset = nullptr;
    // one trailing row: the bookkeeping of the last step stages "step T" (never run)
    if (dalloc(m, &m->d_in, (size_t)(T + 1) * B, true) || dalloc(m, &m->d_out, (size_t)(T + 1) * B, true) ||
        dalloc(m, &m->d_reset, (size_t)(T + 1) * B, true) || dalloc(m, &m->d_M, (size_t)T + 1, true))
        return -1;
    HIPCHK(hipMemcpyAsync(m->d_in, in_idx, (size_t)T * B * sizeof(int), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_out, out_idx, (size_t)T * B * sizeof(int), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_reset, reset, (size_t)T * B, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_M, M, (size_t)T * sizeof(int), hipMemcpyHostToDevice, m->stream));
    m->compact_steps.clear();
    if (n_compact > 0) {
        if (!compact_steps || !compact_maps) return fail("compaction arrays missing");
        if (dalloc(m, &m->d_cmaps, (size_t)n_compact * B, false)) return -1;
        HIPCHK(hipMemcpyAsync(m->d_cmaps, compact_maps, (size_t)n_compact * B * sizeof(int), hipMemcpyHostToDevice, m->stream));
        m->compact_steps.assign(compact_steps, compact_steps + n_compact);
    }
    if (T > m->loss_cap) {
        dfree(m, m->d_loss);
        if (dalloc(m, &m->d_loss, (size_t)T)) return -1;
        m->loss_cap = T;
    }
    {
        // ids of the ACTIVE rows must name catalogue rows: the kernels gather / update table rows by them without a bounds check
        // (rows >= M[t] are never read).  One pass over the host arrays, ~10 ms for an RSC15-sized epoch.
        const int nI = m->dm.n_items;
        for (int64_t t = 0; t < T; ++t) {
            if (M[t] < 0 || M[t] > B) return fail("plan M out of range");      // 0 = padding step (multi-rank plans of unequal length)
            const int32_t *pi = in_idx + t * B, *po = out_idx + t * B;
            unsigned bad = 0;
            for (int b = 0; b < M[t]; ++b) bad |= (unsigned)((unsigned)pi[b] >= (unsigned)nI) | (unsigned)((unsigned)po[b] >= (unsigned)nI);
            if (bad) return fail("plan: item id outside [0, n_items) in an active row of step " + std::to_string(t));
        }
    }
    m->T = T;
    m->dm.in_idx = m->d_in; m->dm.out_idx = m->d_out; m->dm.reset = m->d_reset; m->dm.Mplan = m->d_M;
    m->dm.loss_steps = m->d_loss;
    // the captured graph stays valid: kernels read the plan pointers from the device descriptor
    if (sync_dm(m)) return -1;
    // capture + instantiate the step graph now (capturing executes nothing): the first timed steps of a short run must not
    // pay the ~10 ms of graph construction
    if (m->cfg.use_graph && !m->profiling && !getenv("G4R_TRACE") && (m->dm.apply_dense_inplace || m->comm_ready || m->p2p_ready)) {
        bool whole = false;
        if (ensure_step_graph(m, &whole)) return -1;
        hipGraphExec_t ge = whole ? m->gexec : m->gexec_head;
        if (ge) (void)hipGraphUpload(ge, m->stream);
        if (whole && m->gexec_small) (void)hipGraphUpload(m->gexec_small, m->stream);
    }
    return 0;
}
