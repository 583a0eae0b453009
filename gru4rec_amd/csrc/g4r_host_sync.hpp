// g4r_host_sync.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: reconciliation of the GPU-local item tables (g4r_sync_*, g4r_comm_sync_sparse, virtual ranks).
// ---- reconciliation of the GPU-local item tables ------------------------------------------------------------
static inline int nblk256(long long n) { return (int)((n + 255) / 256); }

int g4r_sync_enable(g4r_model* m) {
    if (!m) return fail("null model");
    if (m->sync_on) return 0;
    if (m->exact) return 0;      // exact-replica mode: the item tables never diverge -- no touched-row bitmap, no base copies
    HIPCHK(hipSetDevice(m->cfg.device));
    DevModel& d = m->dm;
    const size_t I = d.n_items;
    const int tables = d.E ? 2 : 1;
    auto add = [&](int g, float* cur, int W, int kind) -> int {
        if (!cur) return 0;
        float* base = nullptr;
        if (dalloc(m, &base, I * (size_t)W, false)) return -1;
        if (hipMemcpyAsync(base, cur, I * (size_t)W * sizeof(float), hipMemcpyDeviceToDevice, m->stream) != hipSuccess) return fail("base snapshot");
        m->planes[g].push_back({cur, base, W, kind});
        return 0;
    };
    if (add(0, d.Wy, d.Dtop, 0) || add(0, d.accWy, d.Dtop, 1) || add(0, d.velWy, d.Dtop, 0) || add(0, d.acc2Wy, d.Dtop, 1) || add(0, d.cntWy, d.Dtop, 1) ||
        add(0, d.By, 1, 0) || add(0, d.accBy, 1, 1) || add(0, d.velBy, 1, 0) || add(0, d.acc2By, 1, 1) || add(0, d.cntBy, 1, 1))
        return -1;
    if (d.E && (add(1, d.E, d.Ein, 0) || add(1, d.accE, d.Ein, 1) || add(1, d.velE, d.Ein, 0) || add(1, d.acc2E, d.Ein, 1) || add(1, d.cntE, d.Ein, 1))) return -1;
    if (dalloc(m, &m->d_touched, (size_t)tables * I, true) || dalloc(m, &m->d_rowcnt, I, true)) return -1;
    // small item tables: the dense, all-device form of the reconciliation (one all-reduce of [n_items][widths + 1] per table group)
    for (int g = 0; g < tables; ++g) {
        size_t w = 1;
        for (auto& pl : m->planes[g]) w += pl.W;
        const size_t bytes = I * w * sizeof(float);
        if (m->planes[g].size() <= 12 && bytes <= (size_t)env_int("G4R_SYNC_DENSE_MB", 64) * 1024 * 1024 && env_int("G4R_SYNC_DENSE", 1))
            if (dalloc(m, &m->d_dense[g], I * w, false)) return -1;
    }
    // Rule of the optimizer-statistic planes: SUM is right for Adagrad only -- its accumulator is a plain sum of squared
    // gradients, so the ranks' increments add up exactly as they would on one GPU.  rmsprop / adadelta / adam keep MOVING
    // AVERAGES (a <- v a + (1 - v) g^2, gru4rec.py:300-381): each rank's delta contains -(1 - v^k) a0, and the sum over N ranks
    // leaves a0 (1 - N (1 - v^k)) + ... -- negative for rows several ranks touched (v = 0.95, 8 ranks, 16 steps: -3.5 a0), i.e. a
    // NaN in the next sqrt; Adam's first moment would be inflated up to N-fold.  Those statistics take the MEAN over the touching
    // ranks (an average of averages stays inside the range of its inputs), like parameters and velocities.
    if (!m->sync_rule_user) m->sync_rule[1] = (m->cfg.adapt == G4R_ADAPT_ADAGRAD) ? G4R_SYNC_SUM : G4R_SYNC_MEAN;
    if (const char* e = getenv("G4R_SYNC_RULE")) {      // "<param><stat>", s = sum, m = mean: experiments (tools/virtual_ranks_study.py)
        if (e[0]) m->sync_rule[0] = e[0] == 's' ? G4R_SYNC_SUM : G4R_SYNC_MEAN;
        if (e[0] && e[1]) m->sync_rule[1] = e[1] == 's' ? G4R_SYNC_SUM : G4R_SYNC_MEAN;
    }
    d.touched = m->d_touched;
    m->sync_on = true;
    return sync_dm(m);
}

// sorted ids of the rows of `group` this rank rewrote since the last reconciliation
// grow-only scratch (device, or pinned host memory): 0 / -1
static int scratch_ensure(g4r_model::Scratch& sc, size_t bytes, bool host = false) {
    if (bytes == 0) bytes = 16;
    if (sc.p && sc.cap >= bytes) return 0;
    if (sc.p) { if (sc.host) (void)hipHostFree(sc.p); else (void)hipFree(sc.p); sc.p = nullptr; sc.cap = 0; }
    const size_t want = bytes + bytes / 4;      // headroom: the touched set grows and shrinks from call to call
    sc.host = host;
    if (host) HIPCHK(hipHostMalloc(&sc.p, want, hipHostMallocDefault));
    else HIPCHK(hipMalloc(&sc.p, want));
    sc.cap = want;
    return 0;
}
// the rows this rank rewrote since the last reconciliation, as a sorted id list ON THE DEVICE (m->sc_ids): the touched bitmap is
// compacted there (k_touched_count / _scan / _write); only the count comes back
static int sync_local_ids_dev(g4r_model* m, int group, long long* n_out) {
    const long long I = m->dm.n_items;
    const int nb = (int)cdiv(I, TC_CHUNK);
    hipStream_t s = m->stream;
    if (scratch_ensure(m->sc_blk, (size_t)(2 * nb + 2) * sizeof(int))) return -1;
    int* d_blk = (int*)m->sc_blk.p;
    int* d_off = d_blk + nb;
    const unsigned char* t = m->d_touched + (size_t)group * I;
    hipLaunchKernelGGL(k_touched_count, dim3(nb), dim3(256), 0, s, t, I, d_blk);
    hipLaunchKernelGGL(k_touched_scan, dim3(1), dim3(1024), 0, s, (const int*)d_blk, nb, d_off);
    int total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_off + nb, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (scratch_ensure(m->sc_ids, (size_t)std::max(total, 1) * sizeof(int))) return -1;
    if (total > 0) hipLaunchKernelGGL(k_touched_write, dim3(nb), dim3(256), 0, s, t, I, (const int*)d_off, (int*)m->sc_ids.p);
    HIPCHK(hipGetLastError());
    *n_out = total;
    return 0;
}
static int sync_local_ids(g4r_model* m, int group, std::vector<int>& ids) {
    long long n = 0;
    if (sync_local_ids_dev(m, group, &n)) return -1;
    ids.resize((size_t)n);
    if (n > 0) HIPCHK(hipMemcpyAsync(ids.data(), m->sc_ids.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
int64_t g4r_sync_row_floats(g4r_model* m, int32_t group) {
    if (!m || group < 0 || group > 1) { fail("bad argument"); return -1; }
    int64_t w = 0;
    for (auto& pl : m->planes[group]) w += pl.W;
    return w;
}
// test hook / building block: this rank's part = (sorted ids, per plane the delta rows [n][W_p], planes back to back)
int64_t g4r_sync_export(g4r_model* m, int32_t group, int32_t* ids_out, float* rows_out, int64_t cap_rows) {
    if (!m || group < 0 || group > 1) { fail("bad argument"); return -1; }
    if (!m->sync_on) { fail("g4r_sync_enable first"); return -1; }
    if (hipSetDevice(m->cfg.device) != hipSuccess) { fail("hipSetDevice"); return -1; }
    std::vector<int> ids;
    if (sync_local_ids(m, group, ids)) return -1;
    const int64_t n = (int64_t)ids.size();
    if (!ids_out && !rows_out) return n;
    if (n > cap_rows) { fail("export buffers too small"); return -1; }
    if (ids_out) memcpy(ids_out, ids.data(), n * sizeof(int));
    if (rows_out && n > 0) {
        int* d_ids = nullptr; float* d_out = nullptr;
        int wmax = 1;
        for (auto& pl : m->planes[group]) wmax = std::max(wmax, pl.W);
        if (hipMalloc((void**)&d_ids, n * sizeof(int)) != hipSuccess || hipMalloc((void**)&d_out, (size_t)n * wmax * sizeof(float)) != hipSuccess) {
            (void)hipFree(d_ids); fail("export scratch"); return -1;
        }
        (void)hipMemcpyAsync(d_ids, ids.data(), n * sizeof(int), hipMemcpyHostToDevice, m->stream);
        float* dst = rows_out;
        for (auto& pl : m->planes[group]) {
            hipLaunchKernelGGL(k_sync_pack, dim3(nblk256(n * pl.W)), dim3(256), 0, m->stream, (const float*)pl.cur, (const float*)pl.base, pl.W,
                               (const int*)d_ids, (long long)n, d_out);
            (void)hipMemcpyAsync(dst, d_out, (size_t)n * pl.W * sizeof(float), hipMemcpyDeviceToHost, m->stream);
            (void)hipStreamSynchronize(m->stream);
            dst += (size_t)n * pl.W;
        }
        (void)hipFree(d_ids); (void)hipFree(d_out);
        if (hipGetLastError() != hipSuccess) { fail("export kernels"); return -1; }
    }
    return n;
}
// rows of this rank in [lo, hi) of its own sorted list `d_loc` go back to the base, then every part (rank order) is added and
// the rows of every part become the new base.  All pointers are device pointers; part q has cnt[q] rows.
// rowcnt (sync_count below) holds, for the rows of these parts, the number of parts each row occurs in
static void sync_apply(g4r_model* m, const g4r_model::SyncPlane& pl, const int* d_loc, long long n_loc, int nparts,
                       const int* const* d_ids, const long long* cnt, const float* const* d_delta) {
    hipStream_t s = m->stream;
    const unsigned char* rc = (m->sync_rule[pl.kind] == G4R_SYNC_MEAN) ? m->d_rowcnt : nullptr;
    if (n_loc > 0) hipLaunchKernelGGL(k_sync_reset, dim3(nblk256(n_loc * pl.W)), dim3(256), 0, s, pl.cur, (const float*)pl.base, pl.W, d_loc, n_loc);
    for (int q = 0; q < nparts; ++q)
        if (cnt[q] > 0) hipLaunchKernelGGL(k_sync_add, dim3(nblk256(cnt[q] * pl.W)), dim3(256), 0, s, pl.cur, pl.W, d_ids[q], cnt[q], d_delta[q], rc);
    for (int q = 0; q < nparts; ++q)
        if (cnt[q] > 0) hipLaunchKernelGGL(k_sync_rebase, dim3(nblk256(cnt[q] * pl.W)), dim3(256), 0, s, (const float*)pl.cur, pl.base, pl.W, d_ids[q], cnt[q]);
}
// rows-per-part counts of a set of parts (clear = 1: back to zero, after every plane has been applied)
static void sync_count(g4r_model* m, int nparts, const int* const* d_ids, const long long* cnt, int clear) {
    for (int q = 0; q < nparts; ++q)
        if (cnt[q] > 0) hipLaunchKernelGGL(k_sync_count, dim3(nblk256(cnt[q])), dim3(256), 0, m->stream, m->d_rowcnt, d_ids[q], cnt[q], clear);
}
int g4r_sync_set_rule(g4r_model* m, int32_t param_rule, int32_t stat_rule) {
    if (!m || param_rule < 0 || param_rule > G4R_SYNC_MEAN || stat_rule < 0 || stat_rule > G4R_SYNC_MEAN) return fail("bad argument");
    m->sync_rule[0] = param_rule; m->sync_rule[1] = stat_rule; m->sync_rule_user = true;
    return 0;
}
// test hook: apply the parts of all ranks (in rank order; this rank's own part included) as g4r_comm_sync_sparse does after its
// all-gather.  ids[q]: counts[q] sorted item ids; rows[q]: g4r_sync_export layout.
int g4r_sync_import(g4r_model* m, int32_t group, int32_t nparts, const int64_t* counts, const int32_t* const* ids, const float* const* rows) {
    if (!m || group < 0 || group > 1 || nparts < 1 || !counts || !ids || !rows) return fail("bad argument");
    if (!m->sync_on) return fail("g4r_sync_enable first");
    HIPCHK(hipSetDevice(m->cfg.device));
    std::vector<int> loc;
    if (sync_local_ids(m, group, loc)) return -1;
    const size_t I = m->dm.n_items;
    std::vector<int*> d_ids(nparts, nullptr);
    std::vector<float*> d_rows(nparts, nullptr);
    std::vector<long long> cnt(nparts);
    int* d_loc = nullptr;
    const int64_t wsum = g4r_sync_row_floats(m, group);
    auto cleanup = [&]() { for (auto p : d_ids) (void)hipFree(p); for (auto p : d_rows) (void)hipFree(p); (void)hipFree(d_loc); };
    if (!loc.empty()) {
        if (hipMalloc((void**)&d_loc, loc.size() * sizeof(int)) != hipSuccess) { cleanup(); return fail("import scratch"); }
        (void)hipMemcpyAsync(d_loc, loc.data(), loc.size() * sizeof(int), hipMemcpyHostToDevice, m->stream);
    }
    for (int q = 0; q < nparts; ++q) {
        cnt[q] = counts[q];
        if (cnt[q] <= 0) continue;
        for (int64_t j = 0; j < cnt[q]; ++j)
            if (ids[q][j] < 0 || (size_t)ids[q][j] >= I || (j > 0 && ids[q][j] <= ids[q][j - 1])) { cleanup(); return fail("part ids must be sorted, distinct and in range"); }
        if (hipMalloc((void**)&d_ids[q], cnt[q] * sizeof(int)) != hipSuccess || hipMalloc((void**)&d_rows[q], (size_t)cnt[q] * wsum * sizeof(float)) != hipSuccess) {
            cleanup(); return fail("import scratch");
        }
        (void)hipMemcpyAsync(d_ids[q], ids[q], cnt[q] * sizeof(int), hipMemcpyHostToDevice, m->stream);
        (void)hipMemcpyAsync(d_rows[q], rows[q], (size_t)cnt[q] * wsum * sizeof(float), hipMemcpyHostToDevice, m->stream);
    }
    std::vector<const float*> dl(nparts);
    std::vector<size_t> off(nparts, 0);
    sync_count(m, nparts, (const int* const*)d_ids.data(), cnt.data(), 0);
    for (auto& pl : m->planes[group]) {
        for (int q = 0; q < nparts; ++q) dl[q] = d_rows[q] ? d_rows[q] + off[q] : nullptr;
        sync_apply(m, pl, d_loc, (long long)loc.size(), nparts, (const int* const*)d_ids.data(), cnt.data(), dl.data());
        for (int q = 0; q < nparts; ++q) off[q] += (size_t)std::max<long long>(cnt[q], 0) * pl.W;
    }
    sync_count(m, nparts, (const int* const*)d_ids.data(), cnt.data(), 1);
    (void)hipMemsetAsync(m->d_touched + (size_t)group * I, 0, I, m->stream);
    hipError_t e = hipStreamSynchronize(m->stream);
    cleanup();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return fail("import kernels");
    return 0;
}

static SyncPlanes sync_planes_of(g4r_model* m, int group) {
    SyncPlanes p;
    memset(&p, 0, sizeof(p));
    int off = 0;
    for (auto& pl : m->planes[group]) {
        p.cur[p.n] = pl.cur; p.base[p.n] = pl.base; p.W[p.n] = pl.W; p.off[p.n] = off; p.mean[p.n] = m->sync_rule[pl.kind] == G4R_SYNC_MEAN;
        off += pl.W; ++p.n;
    }
    p.wsum = off;
    return p;
}
static void sync_dense_pack(g4r_model* m, int group) {
    const SyncPlanes p = sync_planes_of(m, group);
    const long long I = m->dm.n_items, n = I * (p.wsum + 1);
    hipLaunchKernelGGL(k_sync_dense_pack, dim3(nblk256(n)), dim3(256), 0, m->stream, p, (const unsigned char*)(m->d_touched + (size_t)group * I), I, m->d_dense[group]);
}
static void sync_dense_apply(g4r_model* m, int group) {
    const SyncPlanes p = sync_planes_of(m, group);
    const long long I = m->dm.n_items, n = I * (p.wsum + 1);
    hipLaunchKernelGGL(k_sync_dense_apply, dim3(nblk256(n)), dim3(256), 0, m->stream, p, m->d_touched + (size_t)group * I, I, (const float*)m->d_dense[group]);
}
// The dense reconciliation with the ranks' buffers summed in process (handles of one device standing in for ranks, as in
// g4r_virtual_train_steps): what g4r_comm_sync_sparse does around its ncclAllReduce when the item tables are small.
int g4r_virtual_sync_dense(g4r_model* const* ms, int32_t n) {
    if (!ms || n < 1 || n > 16) return fail("virtual ranks: 1..16 handles");
    for (int q = 0; q < n; ++q) if (!ms[q] || !ms[q]->sync_on || !ms[q]->d_dense[0]) return fail("virtual dense sync: g4r_sync_enable first (and a table small enough for the dense form)");
    HIPCHK(hipSetDevice(ms[0]->cfg.device));
    for (int g = 0; g < 2; ++g) {
        if (ms[0]->planes[g].empty()) continue;
        if (!ms[0]->d_dense[g]) return fail("virtual dense sync: table group too large for the dense form");
        const SyncPlanes p = sync_planes_of(ms[0], g);
        const long long cnt = (long long)ms[0]->dm.n_items * (p.wsum + 1);
        if (cnt > 0x7fffffffLL) return fail("virtual dense sync: buffer too large");
        VSumArgs va;
        memset(&va, 0, sizeof(va));
        for (int q = 0; q < n; ++q) { sync_dense_pack(ms[q], g); va.src[q] = ms[q]->d_dense[g]; va.dst[q] = ms[q]->d_dense[g]; }
        for (int q = 0; q < n; ++q) HIPCHK(hipStreamSynchronize(ms[q]->stream));
        float* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, (size_t)cnt * sizeof(float)));
        hipLaunchKernelGGL(k_virtual_sum, dim3(nblk256(cnt)), dim3(256), 0, ms[0]->stream, va, n, (int)cnt, tmp);
        hipLaunchKernelGGL(k_virtual_bcast, dim3(nblk256(cnt)), dim3(256), 0, ms[0]->stream, va, n, (int)cnt, (const float*)tmp);
        HIPCHK(hipStreamSynchronize(ms[0]->stream));
        (void)hipFree(tmp);
        for (int q = 0; q < n; ++q) sync_dense_apply(ms[q], g);
        for (int q = 0; q < n; ++q) HIPCHK(hipStreamSynchronize(ms[q]->stream));
    }
    return 0;
}

// every table group of this model takes the dense form
static bool sync_all_dense(const g4r_model* m) {
    if (!m->sync_on) return false;
    for (int g = 0; g < 2; ++g) if (!m->planes[g].empty() && !m->d_dense[g]) return false;
    return true;
}
// pack -> all-reduce -> apply for every group, enqueued on the model's stream (no host synchronisation)
static int sync_dense_enqueue(g4r_model* m) {
    for (int g = 0; g < 2; ++g) {
        if (m->planes[g].empty()) continue;
        const SyncPlanes p = sync_planes_of(m, g);
        sync_dense_pack(m, g);
        NCCLCHK(ncclAllReduce(m->d_dense[g], m->d_dense[g], (size_t)m->dm.n_items * (p.wsum + 1), ncclFloat, ncclSum, m->comm, m->stream));
        sync_dense_apply(m, g);
    }
    m->since_sync = 0;
    return 0;
}
// k > 0: g4r_train_steps itself reconciles the item tables every k steps (counted across calls), between two steps, without leaving
// the stream -- only where every table takes the dense form and a communicator exists.  Returns 1 when accepted, 0 when the caller has
// to call g4r_comm_sync_sparse itself (large tables), < 0 on error.  k = 0 switches it off.
int g4r_set_sync_every(g4r_model* m, int32_t k) {
    if (!m || k < 0) return fail("bad argument");
    m->sync_every_dev = 0;
    if (k == 0) return 0;
    if (!m->comm_ready || !sync_all_dense(m)) return 0;
    m->sync_every_dev = k;
    return 1;
}

// RCCL path: id lists all-gathered once per group, then the table is walked in item-id ranges; per range every rank packs its
// delta rows, one all-gather (padded to the largest part of the range) brings all parts, sync_apply adds them in rank order.
// The traffic follows the number of touched rows, not the table size.
int g4r_comm_sync_sparse(g4r_model* m) {
    if (m && m->exact) return 0;      // exact-replica mode: nothing to reconcile
    if (!m) return fail("null model");
    if (m->cfg.nranks <= 1 && !m->comm_ready) return 0;
    if (!m->comm_ready) return fail("g4r_comm_init first");
    if (!m->sync_on) return fail("g4r_sync_enable first");
    HIPCHK(hipSetDevice(m->cfg.device));
    DevModel& d = m->dm;
    int nr = 1;
    NCCLCHK(ncclCommCount(m->comm, &nr));
    const int me = m->cfg.rank;
    const size_t I = d.n_items;
    hipStream_t s = m->stream;
    for (int group = 0; group < 2; ++group) {
        if (m->planes[group].empty()) continue;
        if (m->d_dense[group]) {
            // small table: pack -> one all-reduce -> apply, all on the stream, no host round trip (the sum's order is RCCL's: every
            // rank receives the same bits, so the replicas still end bit-identical)
            const SyncPlanes p = sync_planes_of(m, group);
            sync_dense_pack(m, group);
            NCCLCHK(ncclAllReduce(m->d_dense[group], m->d_dense[group], (size_t)I * (p.wsum + 1), ncclFloat, ncclSum, m->comm, s));
            sync_dense_apply(m, group);
            continue;
        }
        long long mine = 0;
        if (sync_local_ids_dev(m, group, &mine)) return -1;      // sorted ids of this rank's rows in m->sc_ids (device)
        // counts
        std::vector<long long> cnt(nr, 0);
        if (scratch_ensure(m->sc_cnt, (size_t)(nr + 1) * sizeof(long long))) return -1;
        long long* d_cnt = (long long*)m->sc_cnt.p;
        HIPCHK(hipMemcpyAsync(d_cnt + nr, &mine, sizeof(long long), hipMemcpyHostToDevice, s));
        ncclResult_t r = ncclAllGather(d_cnt + nr, d_cnt, 1, ncclInt64, m->comm, s);
        if (r != ncclSuccess) return fail(std::string("ncclAllGather: ") + ncclGetErrorString(r));
        HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nr * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const long long maxn = *std::max_element(cnt.begin(), cnt.end());
        if (maxn == 0) continue;
        // id lists: [nr][maxn], padded with INT_MAX so that every list stays sorted; the host keeps a (pinned) copy for the range walk
        if (scratch_ensure(m->sc_all, (size_t)nr * maxn * sizeof(int)) || scratch_ensure(m->sc_send, (size_t)maxn * sizeof(int)) ||
            scratch_ensure(m->sc_hall, (size_t)nr * maxn * sizeof(int), true)) return -1;
        int *d_all = (int*)m->sc_all.p, *d_send = (int*)m->sc_send.p;
        if (mine < maxn) hipLaunchKernelGGL(k_fill_i32, dim3(nblk256(maxn - mine)), dim3(256), 0, s, d_send + mine, maxn - mine, 0x7fffffff);
        if (mine > 0) HIPCHK(hipMemcpyAsync(d_send, m->sc_ids.p, (size_t)mine * sizeof(int), hipMemcpyDeviceToDevice, s));
        r = ncclAllGather(d_send, d_all, (size_t)maxn, ncclInt32, m->comm, s);
        const int* all = (const int*)m->sc_hall.p;
        if (r != ncclSuccess || hipMemcpyAsync(m->sc_hall.p, d_all, (size_t)nr * maxn * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return fail("id list all-gather failed");
        int wmax = 1;
        for (auto& pl : m->planes[group]) wmax = std::max(wmax, pl.W);
        // item-id ranges: at most `cap` rows per rank and range (bounds the scratch: nr * cap * wmax floats <= ~1 GiB)
        const long long cap = std::max<long long>(1024, (1LL << 28) / ((long long)nr * wmax));
        const long long rows_cap = std::min<long long>(cap, maxn);
        if (scratch_ensure(m->sc_pack, (size_t)rows_cap * wmax * sizeof(float)) || scratch_ensure(m->sc_recv, (size_t)nr * rows_cap * wmax * sizeof(float))) return -1;
        float *d_pack = (float*)m->sc_pack.p, *d_recv = (float*)m->sc_recv.p;
        std::vector<long long> lo(nr, 0), hi(nr, 0), c(nr);
        std::vector<const int*> pid(nr);
        std::vector<const float*> pdl(nr);
        bool ok = true;
        for (long long i0 = 0; i0 < (long long)I && ok;) {
            // the largest id range [i0, i1) in which no rank has more than `cap` rows
            long long i1 = (long long)I;
            for (int q = 0; q < nr; ++q)
                if (lo[q] + cap < cnt[q]) i1 = std::min<long long>(i1, all[(size_t)q * maxn + lo[q] + cap]);
            long long cmax = 0;
            for (int q = 0; q < nr; ++q) {
                const int* b = all + (size_t)q * maxn;
                hi[q] = std::lower_bound(b + lo[q], b + cnt[q], (int)std::min<long long>(i1, 0x7fffffffLL)) - b;
                if (i1 >= (long long)I) hi[q] = cnt[q];
                c[q] = hi[q] - lo[q];
                cmax = std::max(cmax, c[q]);
                pid[q] = d_all + (size_t)q * maxn + lo[q];
            }
            if (cmax > 0) {
                sync_count(m, nr, pid.data(), c.data(), 0);
                for (auto& pl : m->planes[group]) {
                    if (c[me] > 0)
                        hipLaunchKernelGGL(k_sync_pack, dim3(nblk256(c[me] * pl.W)), dim3(256), 0, s, (const float*)pl.cur, (const float*)pl.base, pl.W,
                                           pid[me], c[me], d_pack);
                    if (ncclAllGather(d_pack, d_recv, (size_t)cmax * pl.W, ncclFloat, m->comm, s) != ncclSuccess) { ok = false; break; }
                    for (int q = 0; q < nr; ++q) pdl[q] = d_recv + (size_t)q * cmax * pl.W;
                    sync_apply(m, pl, pid[me], c[me], nr, pid.data(), c.data(), pdl.data());
                }
                sync_count(m, nr, pid.data(), c.data(), 1);
                if (hipStreamSynchronize(s) != hipSuccess) ok = false;
            }
            for (int q = 0; q < nr; ++q) lo[q] = hi[q];
            i0 = i1;
        }
        if (!ok || hipGetLastError() != hipSuccess) return fail("sparse reconciliation failed");
        HIPCHK(hipMemsetAsync(m->d_touched + (size_t)group * I, 0, I, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    m->since_sync = 0;
    return 0;
}
