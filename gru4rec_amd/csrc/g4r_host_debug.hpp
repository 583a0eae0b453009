// g4r_host_debug.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: g4r_get_debug, the row gather / scatter micro-benchmark, the stress load of the asm-pipeline test, the MFMA self-test.
// ------------------------------------------------------------------------------------------------ debug
int g4r_get_debug(g4r_model* m, const char* name, float* host, int64_t count) {
    if (!m || !name || !host) return fail("null argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    DevModel& d = m->dm;
    std::string s(name);
    const float* p = nullptr; int64_t n = 0;
    int l = 0;
    if (!s.empty() && isdigit((unsigned char)s.back())) { l = s.back() - '0'; s.pop_back(); }
    if (l >= d.n_layers) return fail("layer out of range");
    const int64_t bd = (int64_t)d.B * d.D[l];
    if (s == "scores") { p = d.Sc; n = (int64_t)d.B * d.ldSc; }
    // (step planes: the ring slot of the last step run)
    else if (s == "dSx") { p = d.dSx + (size_t)((m->gstep - 1) & d.defer_mask) * (size_t)d.dSx_stride; n = (int64_t)d.B * d.Ein; }
    else if (s == "dSy") { p = d.dSy + (size_t)((m->gstep - 1) & d.defer_mask) * (size_t)d.dSy_stride; n = (int64_t)d.ldSc * d.Dtop; }
    else if (s == "dSBy") { p = d.dSBy + (size_t)((m->gstep - 1) & d.defer_mask) * (size_t)d.dSBy_stride; n = d.ldSc; }
    else if (s == "defer_stats") {      // (rows applied by flush launches, bias entries, 1 if deferral is on, slots)
        if (count < 4) return fail("count");
        double rows = 0, bias = 0;
        if (m->defer_on) {
            std::vector<unsigned> st(2048);
            HIPCHK(hipStreamSynchronize(m->stream));
            HIPCHK(hipMemcpy(st.data(), d.dstat, st.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < st.size(); i += 2) { rows += st[i]; bias += st[i + 1]; }
        }
        host[0] = (float)rows; host[1] = (float)bias; host[2] = m->defer_on ? 1.f : 0.f; host[3] = (float)(d.defer_mask + 1);
        return 0;
    }
    else if (s == "dhpart") { p = d.dhpart; n = (int64_t)d.ksplit * d.B * d.Dtop; }
    else if (s == "lossrow") { p = d.lossrow; n = d.B; }
    else if (s == "hd") { p = d.hd[l]; n = bd; }
    else if (s == "r") { p = d.r[l]; n = bd; }
    else if (s == "z") { p = d.z[l]; n = bd; }
    else if (s == "c") { p = d.c[l]; n = bd; }
    else if (s == "Hr") { p = d.Hr[l]; n = bd; }
    else if (s == "dV") { p = d.dV[l]; n = bd * 3; }
    else if (s == "dyl") { p = d.dyl[l]; n = bd; }
    else if (s == "Hprev") { p = d.H[l][(m->gstep + 1) & 1]; n = bd; }
    else if (s == "occ_idx") { p = (const float*)d.occ_idx; n = d.R; }
#if !defined(G4R_CLK_TRACE)
    else if (s == "dbgclk" || s == "dbgtile") return fail("in-kernel traces need a library built with G4R_BUILD_CLK=1 (python -m gru4rec_amd.build --force) and G4R_CLK=1 at run time");
#endif
    else if (s == "dbgclk") { if (!d.dbgclk) return fail("G4R_CLK not set"); p = (const float*)d.dbgclk; n = 2 * (64 + 8 * (int64_t)d.R); }
    else if (s == "dbgtile") { if (!d.dbgtile) return fail("G4R_CLK not set"); p = (const float*)d.dbgtile; n = 2 * 8 * (int64_t)8192; }      // [0, 4096): dense tiles, [4096, 8192): k_score_fwd tiles
    else if (s == "ntiles") { if (count < 1) return fail("count"); host[0] = (float)m->ntiles; return 0; }
    else if (s == "ldSc") { if (count < 1) return fail("count"); host[0] = (float)d.ldSc; return 0; }
    else if (s == "wide_mask") {      // which wide-layer kernels run (bits 1 / 2 / 4 / 8 per layer OR-ed, 16 = k_dense_grad2)
        if (count < 1) return fail("count");
        int mk = m->wide_dense ? 16 : 0;
        for (int l = 0; l < d.n_layers; ++l) mk |= m->wg[l].use;
        host[0] = (float)mk; return 0;
    }
    else if (s == "deep_geo") {      // layer 0 at the training batch: 1 = k_gru_p2 on 8 waves x 256-deep chunks, 2 = k_gru_bwd_a (deep_geometry)
        if (count < 1) return fail("count");
        host[0] = (float)(deep_geometry(m->p2_geo_env, m->n_cu, d.D[0], d.B) + 2 * deep_geometry(m->ba_geo_env, m->n_cu, d.D[0], d.B)); return 0;
    }
    else if (s == "score_mt") { if (count < 1) return fail("count"); host[0] = (float)score_mt_width(d, m->n_cu); return 0; }      // macro-tile width of the scoring forward (0: 64 x 64 tiles)
    else if (s == "score_bmt") { if (count < 1) return fail("count"); host[0] = (float)score_bmt_slabs(d, m->n_cu); return 0; }      // slabs of the macro-tile scoring backward (0: k_score_bwd2)
    else if (s == "ksplit") { if (count < 1) return fail("count"); host[0] = (float)d.ksplit; return 0; }
    else if (s == "dev_syncs") { if (count < 1) return fail("count"); host[0] = (float)m->n_dev_syncs; return 0; }
    else if (s == "dense_count") { if (count < 1) return fail("count"); host[0] = (float)d.dense_count; return 0; }
    else if (s == "occ_score_tile") {      // resident workgroups per CU the runtime reports for the gemm_tile2 scoring kernel
        if (count < 1) return fail("count");
        int nb = 0;
        for (size_t lds = SMEM_SF2; lds >= SMEM_SF2 - 2048; lds -= 512) {
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_score_fwd_t2, GT_NTH, lds));
            fprintf(stderr, "[g4r] k_score_fwd_t2 dynamic LDS %zu -> %d workgroups per CU\n", lds, nb);
        }
        host[0] = (float)nb;
        return 0;
    }
    else if (s == "graph_mode") {      // 0: no graph yet, 1: whole steps replayed (RCCL captured when N > 1), 2: head graph + eager tail
        if (count < 1) return fail("count");
        host[0] = m->gexec ? 1.f : (m->gexec_head ? 2.f : 0.f);
        return 0;
    }
    else return fail(std::string("unknown debug buffer ") + name);
    if (count != n) return fail(std::string("size mismatch for debug buffer ") + name + " expected " + std::to_string(n));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(host, p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---- row gather / scatter micro-benchmark (g4r_micro_kernels.cuh) ------------------------------------------------
int g4r_bench_rows(int32_t device, int64_t n_items, int32_t W, int64_t rows_per_launch, int32_t launches, int32_t mode, uint64_t seed,
                   double* kernel_us, double* wall_us) {
    if (n_items < 1 || W < 4 || W % 4 != 0 || W > 512 || rows_per_launch < 1 || launches < 1 || mode < 0 || mode > 2 || !kernel_us || !wall_us)
        return fail("bad argument");
    if (device < 0 || device >= g4r_device_count()) return fail("device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    float *table = nullptr, *acc = nullptr, *buf = nullptr;
    int* idx = nullptr;
    hipStream_t s = nullptr;
    std::vector<hipEvent_t> ev;
    auto cleanup = [&]() {
        (void)hipFree(table); (void)hipFree(acc); (void)hipFree(buf); (void)hipFree(idx);
        for (auto e : ev) (void)hipEventDestroy(e);
        if (s) (void)hipStreamDestroy(s);
    };
#define MBCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(std::string(#x ": ") + hipGetErrorString(e_)); } } while (0)
    const size_t tab = (size_t)n_items * W;
    const int warm = 3, total = launches + warm;
    MBCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    MBCHK(hipMalloc((void**)&table, tab * sizeof(float)));
    MBCHK(hipMemsetAsync(table, 0, tab * sizeof(float), s));
    if (mode == 2) { MBCHK(hipMalloc((void**)&acc, tab * sizeof(float))); MBCHK(hipMemsetAsync(acc, 0, tab * sizeof(float), s)); }
    MBCHK(hipMalloc((void**)&buf, (size_t)rows_per_launch * W * sizeof(float)));
    MBCHK(hipMemsetAsync(buf, 0, (size_t)rows_per_launch * W * sizeof(float), s));
    // every launch gets its own rows (distinct within a launch: a random start and an odd stride modulo n_items would cluster,
    // so a multiplicative hash of a counter is used; duplicates inside a launch are a fraction ~rows/n_items and harmless here)
    std::vector<int> h((size_t)total * rows_per_launch);
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    for (auto& v : h) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; v = (int)(((x * 0x2545F4914F6CDD1Dull) >> 11) % (unsigned long long)n_items); }
    MBCHK(hipMalloc((void**)&idx, h.size() * sizeof(int)));
    MBCHK(hipMemcpyAsync(idx, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, s));
    MBCHK(hipStreamSynchronize(s));
    ev.resize(2 * (size_t)launches + 2);
    for (auto& e : ev) MBCHK(hipEventCreate(&e));
    const long long waves = (rows_per_launch + MB_RPW - 1) / MB_RPW;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    for (int l = 0; l < total; ++l) {
        const int* ix = idx + (size_t)l * rows_per_launch;
        const int t = l - warm;
        if (t == 0) MBCHK(hipEventRecord(ev[2 * (size_t)launches], s));
        hipEvent_t a = t >= 0 ? ev[2 * (size_t)t] : nullptr, b = t >= 0 ? ev[2 * (size_t)t + 1] : nullptr;
        if (W <= 256) hipExtLaunchKernelGGL(k_micro_rows<1>, grid, block, 0, s, a, b, 0, (const float*)table, acc, ix, buf, (long long)rows_per_launch, (int)W, (int)mode);
        else hipExtLaunchKernelGGL(k_micro_rows<2>, grid, block, 0, s, a, b, 0, (const float*)table, acc, ix, buf, (long long)rows_per_launch, (int)W, (int)mode);
    }
    MBCHK(hipEventRecord(ev[2 * (size_t)launches + 1], s));
    MBCHK(hipStreamSynchronize(s));
    MBCHK(hipGetLastError());
    double ksum = 0.0;
    for (int t = 0; t < launches; ++t) { float ms = 0.f; MBCHK(hipEventElapsedTime(&ms, ev[2 * (size_t)t], ev[2 * (size_t)t + 1])); ksum += ms; }
    float wall = 0.f;
    MBCHK(hipEventElapsedTime(&wall, ev[2 * (size_t)launches], ev[2 * (size_t)launches + 1]));
#undef MBCHK
    *kernel_us = 1000.0 * ksum / launches;
    *wall_us = 1000.0 * wall / launches;
    cleanup();
    return 0;
}

// ---- memory-system load for the stress test (tests/test_gpu_stress.py): `launches` passes of k_stress_stream over `mbytes` MiB on a
// stream of their own, queued asynchronously; g4r_stress_stop waits for them and frees the buffer
struct g4r_stress { int device; float* buf; hipStream_t s; };
int g4r_stress_start(int32_t device, int64_t mbytes, int32_t launches, void** handle) {
    if (!handle || mbytes < 1 || launches < 1 || launches > 4096) return fail("bad argument");
    if (device < 0 || device >= g4r_device_count()) return fail("device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    g4r_stress* h = new g4r_stress{device, nullptr, nullptr};
    const size_t bytes = (size_t)mbytes << 20;
    if (hipMalloc((void**)&h->buf, bytes) != hipSuccess) { delete h; (void)hipGetLastError(); return fail("stress buffer allocation failed"); }
    if (hipStreamCreateWithFlags(&h->s, hipStreamNonBlocking) != hipSuccess) { (void)hipFree(h->buf); delete h; return fail("stress stream"); }
    (void)hipMemsetAsync(h->buf, 0, bytes, h->s);
    const long long n4 = (long long)(bytes / 16);
    const unsigned grid = (unsigned)((n4 + 16383) / 16384);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_stress_stream, dim3(grid), dim3(256), 0, h->s, h->buf, n4);
    *handle = h;
    return 0;
}
int g4r_stress_stop(void* handle) {
    if (!handle) return fail("null handle");
    g4r_stress* h = (g4r_stress*)handle;
    (void)hipSetDevice(h->device);
    hipError_t e = hipStreamSynchronize(h->s);
    (void)hipStreamDestroy(h->s);
    (void)hipFree(h->buf);
    delete h;
    if (e != hipSuccess) return fail(std::string("stress stream: ") + hipGetErrorString(e));
    return 0;
}

int g4r_selftest_mfma(float* max_abs_err) {
    if (g4r_device_count() <= 0) return fail("no HIP device");
    const int K = 20;
    std::vector<float> A(16 * K), Bm(K * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) A[i * K + k] = 0.25f * (float)((i * 7 + k * 3) % 11) - 1.0f;
    for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) Bm[k * 16 + j] = 0.5f * (float)((k * 5 + j * 13) % 9) - 2.0f + 0.01f * j;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], Bm[k * 16 + j], s); R[i * 16 + j] = s; }
    float *dA, *dB, *dC;
    HIPCHK(hipMalloc(&dA, A.size() * 4)); HIPCHK(hipMalloc(&dB, Bm.size() * 4)); HIPCHK(hipMalloc(&dC, 256 * 4));
    HIPCHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, Bm.data(), Bm.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, 0, (const float*)dA, (const float*)dB, dC, K);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    float e = 0.f;
    for (int i = 0; i < 256; ++i) e = std::max(e, std::fabs(C[i] - R[i]));
    // 32x32x2 shape (gemm_tile2)
    {
        const int K2 = 18;
        std::vector<float> A2(32 * K2), B2(K2 * 32), C2(1024), R2(1024, 0.f);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < K2; ++k) A2[i * K2 + k] = 0.25f * (float)((i * 5 + k * 3) % 13) - 1.5f;
        for (int k = 0; k < K2; ++k) for (int j = 0; j < 32; ++j) B2[k * 32 + j] = 0.5f * (float)((k * 7 + j * 11) % 9) - 2.0f + 0.01f * j;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0.f; for (int k = 0; k < K2; ++k) s = fmaf(A2[i * K2 + k], B2[k * 32 + j], s); R2[i * 32 + j] = s; }
        float *dA2, *dB2, *dC2;
        HIPCHK(hipMalloc(&dA2, A2.size() * 4)); HIPCHK(hipMalloc(&dB2, B2.size() * 4)); HIPCHK(hipMalloc(&dC2, 1024 * 4));
        HIPCHK(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dB2, B2.data(), B2.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_selftest_mfma32, dim3(1), dim3(64), 0, 0, (const float*)dA2, (const float*)dB2, dC2, K2);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(C2.data(), dC2, 1024 * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dA2); (void)hipFree(dB2); (void)hipFree(dC2);
        for (int i = 0; i < 1024; ++i) e = std::max(e, std::fabs(C2[i] - R2[i]));
    }
    if (max_abs_err) *max_abs_err = e;
    return 0;
}
