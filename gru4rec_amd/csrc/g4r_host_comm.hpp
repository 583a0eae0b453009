// g4r_host_comm.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: RCCL communicator, reductions of scalars, the one-shot all-reduce through peer memory (g4r_p2p_*).
// ------------------------------------------------------------------------------------------------ RCCL

int g4r_comm_unique_id(char* out128) {
    if (!out128) return fail("null argument");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) <= 128, "unique id size");
    memset(out128, 0, 128);
    memcpy(out128, &id, sizeof(id));
    return 0;
}
int g4r_comm_init(g4r_model* m, const char* id128, int32_t nranks, int32_t rank) {
    if (!m || !id128) return fail("null argument");
    if (nranks != m->cfg.nranks || rank != m->cfg.rank) return fail("rank layout differs from g4r_config");
    HIPCHK(hipSetDevice(m->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    // RCCL prints a version banner on stdout when a communicator is created; stdout belongs to the caller (bench.py prints one
    // JSON line there), so file descriptor 1 points at stderr while RCCL initialises
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) (void)dup2(2, 1);
    const ncclResult_t rc_init = ncclCommInitRank(&m->comm, nranks, id, rank);
    fflush(stdout);
    if (saved_out >= 0) { (void)dup2(saved_out, 1); (void)close(saved_out); }
    NCCLCHK(rc_init);
    m->comm_ready = true;
    return g4r_sync_enable(m);      // base snapshot of the item tables as they are now (g4r_set_param keeps it in step)
}
static int comm_reduce_i64(g4r_model* m, int64_t* value, ncclRedOp_t op) {
    if (!m || !value) return fail("null argument");
    if (m->cfg.nranks <= 1 && !m->comm_ready) return 0;
    if (!m->comm_ready) return fail("g4r_comm_init first");
    HIPCHK(hipSetDevice(m->cfg.device));
    long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(long long)));
    HIPCHK(hipMemcpyAsync(d, value, sizeof(long long), hipMemcpyHostToDevice, m->stream));
    ncclResult_t r = ncclAllReduce(d, d, 1, ncclInt64, op, m->comm, m->stream);
    if (r != ncclSuccess) { (void)hipFree(d); return fail(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
    HIPCHK(hipMemcpyAsync(value, d, sizeof(long long), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    (void)hipFree(d);
    return 0;
}
int g4r_comm_min_i64(g4r_model* m, int64_t* value) { return comm_reduce_i64(m, value, ncclMin); }
int g4r_comm_max_i64(g4r_model* m, int64_t* value) { return comm_reduce_i64(m, value, ncclMax); }
int g4r_comm_nranks(g4r_model* m) {
    if (!m) { fail("null model"); return -1; }
    if (!m->comm_ready) return 1;
    int n = 0;
    if (ncclCommCount(m->comm, &n) != ncclSuccess) { fail("ncclCommCount failed"); return -1; }
    return n;
}

// ---- one-shot all-reduce through peer memory (k_p2p_allreduce, g4r_sync_kernels.cuh) -------------------------------------------
// The switch next to the RCCL all-reduce of the dense gradients: g4r_p2p_enable on a handle that has a communicator (the 64-byte
// IPC handles travel through one ncclAllGather), or g4r_p2p_export / g4r_p2p_attach with the handles carried by the caller (no
// RCCL at all: two processes on ONE device can be ranks of each other that way, which RCCL refuses -- the N > 1 test a one-GPU box
// can run).  One node only: the handles are hipIpcMemHandle_t.
static int p2p_timeout_ms() { const char* e = getenv("G4R_P2P_TIMEOUT_MS"); return e ? std::max(1, atoi(e)) : 20000; }
int g4r_p2p_export(g4r_model* m, char* out_handle64) {
    if (!m || !out_handle64) return fail("null argument");
    if (m->dm.apply_dense_inplace) return fail("p2p: the handle was created as a single rank (nranks = 1 without G4R_FORCE_STAGED)");
    if (m->p2p_ready) return fail("p2p: already attached");
    if (m->cfg.nranks > G4R_P2P_MAX) return fail("p2p: at most 8 ranks (one node)");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    HIPCHK(hipSetDevice(m->cfg.device));
    if (!m->p2p_region) {
        m->p2p_nblk = cdiv(m->dm.dense_count, 1024);
        m->p2p_cap = m->p2p_nblk * 1024;
        const size_t flag_bytes = ((size_t)m->p2p_nblk * sizeof(unsigned) + 4095) & ~(size_t)4095;
        const size_t bytes = flag_bytes + 2 * (size_t)m->p2p_cap * sizeof(float);
        // uncached (fine-grained) device memory where the runtime exports it; plain device memory otherwise -- every access of
        // the kernel is system scope either way
        void* q = nullptr;
        hipIpcMemHandle_t h;
        bool ok = false;
        if (hipExtMallocWithFlags(&q, bytes, hipDeviceMallocUncached) == hipSuccess) {
            ok = hipIpcGetMemHandle(&h, q) == hipSuccess;
            if (!ok) { (void)hipFree(q); q = nullptr; }
        }
        (void)hipGetLastError();
        if (!ok) {
            HIPCHK(hipMalloc(&q, bytes));
            if (hipIpcGetMemHandle(&h, q) != hipSuccess) { (void)hipFree(q); return fail("p2p: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); }
        }
        HIPCHK(hipMemset(q, 0, bytes));
        m->p2p_region = q;
        if (dalloc(m, &m->p2p_round, (size_t)m->p2p_nblk + 1)) return -1;
        HIPCHK(hipStreamSynchronize(m->stream));
        memcpy(out_handle64, &h, 64);
        return 0;
    }
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, m->p2p_region));
    memcpy(out_handle64, &h, 64);
    return 0;
}
int g4r_p2p_attach(g4r_model* m, const char* handles, int32_t nranks, int32_t rank) {
    if (!m || !handles) return fail("null argument");
    if (!m->p2p_region) return fail("p2p: g4r_p2p_export first");
    if (m->p2p_ready) return fail("p2p: already attached");
    if (nranks != m->cfg.nranks || rank != m->cfg.rank) return fail("rank layout differs from g4r_config");
    if (nranks < 1 || nranks > G4R_P2P_MAX) return fail("p2p: 1..8 ranks");
    HIPCHK(hipSetDevice(m->cfg.device));
    const size_t flag_bytes = ((size_t)m->p2p_nblk * sizeof(unsigned) + 4095) & ~(size_t)4095;
    P2PArgs& a = m->p2p_args;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < nranks; ++q) {
        char* base = (char*)m->p2p_region;
        if (q != rank) {
            hipIpcMemHandle_t h;
            memcpy(&h, handles + 64 * (size_t)q, 64);
            void* ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("p2p: hipIpcOpenMemHandle (rank ") + std::to_string(q) + "): " + hipGetErrorString(e)); }
            m->p2p_peer[q] = ptr;
            base = (char*)ptr;
        }
        a.flags[q] = (unsigned*)base;
        a.data[q] = (float*)(base + flag_bytes);
    }
    a.own_flags = a.flags[rank]; a.own_data = a.data[rank];
    a.round = m->p2p_round;
    a.nranks = nranks; a.rank = rank; a.count = m->dm.dense_count; a.cap = m->p2p_cap; a.nblk = m->p2p_nblk;
    a.spin_ticks = (long long)p2p_timeout_ms() * 100000;      // wall_clock64: 100 MHz
    // a step graph captured with the RCCL node is stale now
    if (m->gexec) { (void)hipGraphExecDestroy(m->gexec); m->gexec = nullptr; }
    if (m->gexec_small) { (void)hipGraphExecDestroy(m->gexec_small); m->gexec_small = nullptr; }
    m->p2p_ready = true;
    return 0;
}
int g4r_p2p_enable(g4r_model* m) {
    if (!m) return fail("null model");
    if (!m->comm_ready) return fail("g4r_comm_init first (or carry the handles yourself: g4r_p2p_export / g4r_p2p_attach)");
    const int n = m->cfg.nranks;
    std::vector<char> all(64 * (size_t)n);
    if (g4r_p2p_export(m, all.data() + 64 * (size_t)m->cfg.rank)) return -1;
    char* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 64 * (size_t)n));
    HIPCHK(hipMemcpyAsync(d + 64 * (size_t)m->cfg.rank, all.data() + 64 * (size_t)m->cfg.rank, 64, hipMemcpyHostToDevice, m->stream));
    ncclResult_t r = ncclAllGather(d + 64 * (size_t)m->cfg.rank, d, 64, ncclChar, m->comm, m->stream);
    if (r != ncclSuccess) { (void)hipFree(d); return fail(std::string("ncclAllGather: ") + ncclGetErrorString(r)); }
    HIPCHK(hipMemcpyAsync(all.data(), d, 64 * (size_t)n, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    (void)hipFree(d);
    return g4r_p2p_attach(m, all.data(), n, m->cfg.rank);
}
int g4r_p2p_active(g4r_model* m) { return m && m->p2p_ready ? 1 : 0; }
