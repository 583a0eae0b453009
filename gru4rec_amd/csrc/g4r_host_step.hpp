// g4r_host_step.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: the training step: launch_step (every launch of a step, in order), tail compaction, the captured step graphs, g4r_train_steps (+ virtual ranks), losses, counters, per-kernel profiling.
// ------------------------------------------------------------------------------------------------ the step
static inline bool no_merge_tail() { static const bool v = getenv("G4R_NO_MERGE") != nullptr; return v; }
// float4 chunks per lane a gathered row needs in the sparse update: rows of <= 256 / 512 / 1024 floats
static inline int row_chunks(const DevModel& d) { const int w = std::max(d.Dtop, d.Ein); return w <= 256 ? 1 : (w <= 512 ? 2 : 4); }
// (rows wider than 512 floats take the two-launch form: k_update's register budget is sized for two chunks per lane)
// (wide layers: the dense gradients run as 64 x 64 tiles in a launch of their own, k_dense_grad2, ahead of the sparse row update)
// the lean form of the merged update (k_update_l, g4r_lean_kernels.cuh): one GPU, Adagrad(+momentum) with the dense rule fused, batches of <= 128
// rows, item rows of <= 256 floats, no deferral, no touched-row bitmap (N > 1)
static inline bool lean_update(const g4r_model* m);
static inline bool merged_update(const g4r_model* m) { return !m->dm.generic && !no_merge_tail() && row_chunks(m->dm) <= 2 && !m->wide_dense; }
static inline bool lean_update(const g4r_model* m) {
    static const bool off = getenv("G4R_NO_LEAN") != nullptr;
    const DevModel& d = m->dm;
    return !off && m->lean_upd && merged_update(m) && m->d_leanU && d.apply_dense_inplace && d.B <= 128 && row_chunks(d) == 1 && !m->defer_on && d.xmode == 0 && !d.touched &&
           d.R < 65536 && m->ntiles16 < 65536 && cdiv(d.R, 8) < 65536;
}
// part: 0 = the whole step; 1 = head (everything up to the dense gradients); 2 = tail (all-reduce, dense apply, sparse update)
static int launch_step(g4r_model* m, std::vector<EvRec>* recs, int part = 0) {
    DevModel& d = m->dm;
    const int L = d.n_layers, B = d.B;
    hipStream_t s = m->stream;
    GruFwdPredict nopa = {};
    size_t evi = 0;
    hipEvent_t cur_a = nullptr, cur_b = nullptr;
    auto begin = [&](int kn) {
        if (!recs) return;
        while (m->evs.size() < evi + 2) { hipEvent_t e; (void)hipEventCreate(&e); m->evs.push_back(e); }
        EvRec r = {kn, m->evs[evi], m->evs[evi + 1]};
        evi += 2;
        cur_a = r.a; cur_b = r.b;     // attached to the dispatch itself (hipExtLaunchKernelGGL): kernel-only duration
        recs->push_back(r);
    };
    static const bool trace = getenv("G4R_TRACE") != nullptr;
    // measurement aid (tools/kn_cost.py): bit k set = the launches of kernel slot k (KN_*) are left out of the step -- the step's results are
    // garbage, its duration tells what that launch costs the captured step (HIP events and rocprofv3 both put a ~3 us floor under a dispatch
    // that the graph does not pay: an empty kernel costs 1.5 us there, tools/probes/chain_probe.hip)
    static const unsigned long long skip_kn = getenv("G4R_SKIP_KN") ? strtoull(getenv("G4R_SKIP_KN"), nullptr, 0) : 0ull;
    int trace_kn = -1;
    auto begin0 = begin;
    auto begin_t = [&](int kn) {
        trace_kn = kn;
        if (trace) { fprintf(stderr, "[g4r] launch %s\n", KN_NAMES[kn]); fflush(stderr); }
        begin0(kn);
    };
    auto end = [&]() {
        if (trace) {
            hipError_t e = hipStreamSynchronize(s);
            fprintf(stderr, "[g4r] done   %s: %s\n", KN_NAMES[trace_kn], hipGetErrorString(e));
            fflush(stderr);
        }
    };
#define begin begin_t
#define LK(kern, grid, block, smem, strm, ...)                                                             \
    do {                                                                                                  \
        if (skip_kn && trace_kn >= 0 && ((skip_kn >> trace_kn) & 1ull)) break;                            \
        if (recs) hipExtLaunchKernelGGL(kern, grid, block, smem, strm, cur_a, cur_b, 0, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kern, grid, block, smem, strm, __VA_ARGS__);                              \
    } while (0)
    const DevModel* dmp = (const DevModel*)m->d_dm;
    StepState* stp = (StepState*)d.st;
    bool merged = false;      // the sparse update already ran inside k_update
    if (part != 2) {
    for (int l = 0; l < L; ++l) {
        if (lean_gru(d, l)) {
            const int ntd = cdiv(d.D[l], 16), nrb = cdiv(B, 16);
            begin(KN_GRU_V);
            {
                const LeanV& v = m->h_leanV[l];
                const unsigned dims = (unsigned)d.D[l] | ((unsigned)d.IN[l] << 16);
#define G4R_LK_V(L0_, DR_) LK((k_gru_v<L0_, DR_>), dim3(ntd, nrb, 3), dim3(512), 0, s, (const LeanV*)(m->d_leanV + l), (StepState*)v.st, (const int*)v.cur_in, \
                              (const float*)v.Wx, (const float*)v.Wrz, (const float*)v.H0, (const float*)v.H1, dims, (unsigned)B)
                if (l > 0) G4R_LK_V(false, false); else if (d.drop_e > 0.f) G4R_LK_V(true, true); else G4R_LK_V(true, false);
#undef G4R_LK_V
            }
            end();
            begin(KN_GRU_H);
            {
                const LeanH& h = m->h_leanH[l];
                LK(k_gru_h, dim3(ntd, nrb), dim3(512), 0, s, (const LeanH*)(m->d_leanH + l), (const float*)h.Wh, (const float*)h.Hr,
                   (const float*)h.Vc, (const float*)h.z, (const int*)h.cur_rst, (const float*)h.H0, (const float*)h.H1, (unsigned)d.D[l], (unsigned)B);
            }
            end();
            continue;
        }
        if (fused_fwd(d, l)) {
            begin(KN_FWD_FUSED);
            LK(k_gru_fwd_fused, dim3(cdiv(d.D[l], 32), cdiv(B, FF_ROWS)), dim3(512), (size_t)fwd_fused_lds(d.IN[l], d.D[l]).total * sizeof(float), s, dmp, stp, l, l == 0 ? 1 : 0);
            end();
            continue;
        }
        const g4r_model::WideGeo& G = m->wg[l];
        const int nrt64 = cdiv(B, 64), nct64 = d.D[l] / 64;
        begin(KN_GRU_P1);
        if (G.use & 1) {
            LK(k_gru_p1s, dim3(nct64 * nrt64 * (3 * G.ny + 2 * G.nh)), dim3(256), SMEM_T2K, s, dmp, stp, l, l == 0 ? 1 : 0, G.ny, G.nh, G.kys, G.khs);
            end();
            begin(KN_GATE);
            LK(k_gru_gate, dim3(cdiv((long long)B * (d.D[l] / 4), 256)), dim3(256), 0, s, dmp, stp, l, G.ny, G.nh);
        } else if (wide_layer(d.D[l])) LK(k_gru_p1_n64, dim3(cdiv(3 * d.D[l], 64), cdiv(B, GT_BM)), dim3(GT_NTH_FEW), SMEM_P1_N64, s, dmp, stp, l, 1, l == 0 ? 1 : 0, nopa);
        else LK(k_gru_p1_n32, dim3(cdiv(3 * d.D[l], GT_BN), cdiv(B, GT_BM)), dim3(GT_NTH_FEW), SMEM_P1, s, dmp, stp, l, 1, l == 0 ? 1 : 0, nopa);
        end();
        begin(KN_GRU_P2);
        {
            const dim3 g2(cdiv(d.D[l], GT_BN), cdiv(B, GT_BM));
            if (deep_geometry(m->p2_geo_env, m->n_cu, d.D[l], B)) LK(k_gru_p2_w8d, g2, dim3(512), SMEM_P2_256, s, dmp, stp, l, 1, nopa);
            else LK(k_gru_p2_w4, g2, dim3(GT_NTH), SMEM_NN, s, dmp, stp, l, 1, nopa);
        }
        end();
    }
    begin(KN_SCORE_FWD);
    if (lean_scores(d)) {
        const unsigned dimsA = (unsigned)d.Dtop | ((unsigned)B << 16), dimsB = (unsigned)d.N | ((unsigned)d.ldSc << 16);
        const dim3 gs(cdiv(d.ldSc, 32), cdiv(B, 32));
        if (d.logq != 0.f) LK(k_score_s<true>, gs, dim3(256), 0, s, (const LeanS*)m->d_leanS, (const int*)(d.cur_in + 2 * B), (const int*)d.cur_col, (const float*)d.hd[L - 1],
                              (const float*)d.Wy, (const float*)d.By, (float*)d.Sc, dimsA, dimsB);
        else LK(k_score_s<false>, gs, dim3(256), 0, s, (const LeanS*)m->d_leanS, (const int*)(d.cur_in + 2 * B), (const int*)d.cur_col, (const float*)d.hd[L - 1],
                (const float*)d.Wy, (const float*)d.By, (float*)d.Sc, dimsA, dimsB);
    } else if (score_mt_width(d, m->n_cu) == 272) LK(k_score_mt_4s, dim3(cdiv(B, 64) * cdiv(d.ldSc, 272)), dim3(256), SMEM_MT_4S, s, (const int*)d.cur_col, (const int*)(d.cur_in + 2 * B),
                                                       (const float*)d.hd[L - 1], (const float*)d.Wy, (const float*)d.zrow, dmp, (unsigned)d.Dtop | ((unsigned)cdiv(B, 64) << 16),
                                                       (unsigned)d.N, (unsigned)d.ldSc, (unsigned)B);
    else if (score_fwd_dma(d)) LK(k_score_fwd_t3, dim3(cdiv(d.ldSc, 64), cdiv(B, 64)), dim3(GT_NTH), SMEM_SF3, s, dmp, stp);
    else if (wide_scores(d) && score_tile2() && d.Dtop % T2_BK == 0) LK(k_score_fwd_t2, dim3(cdiv(d.ldSc, 64), cdiv(B, 64)), dim3(GT_NTH), SMEM_SF2, s, dmp, stp);
    else if (wide_scores(d)) LK(k_score_fwd_k64, dim3(cdiv(d.ldSc, SFW_BN), cdiv(B, SF_BM)), dim3(GT_NTH), SMEM_SF64, s, dmp, stp);
    else LK(k_score_fwd_k128, dim3(cdiv(d.ldSc, GT_BN), cdiv(B, SF_BM)), dim3(GT_NTH), SMEM_SF, s, dmp, stp);
    end();
    begin(KN_LOSS);
    {
        // the (final activation, loss) pairs of BASELINE's configurations run compile-time specialised builds of the kernel
        const int spec = (d.final_act == G4R_ACT_ELU && d.loss == G4R_LOSS_BPR_MAX) ? 1
                       : (d.final_act == G4R_ACT_SOFTMAX && d.loss == G4R_LOSS_XE) ? 2
                       : (d.final_act == G4R_ACT_ELU && d.loss == G4R_LOSS_TOP1_MAX) ? 3 : 0;
#define G4R_LK_LOSS(L, V)                                                                              \
        do {                                                                                           \
            if (spec == 1) LK((k_loss_rows<L, 1, V>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp);      \
            else if (spec == 2) LK((k_loss_rows<L, 2, V>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp); \
            else if (spec == 3) LK((k_loss_rows<L, 3, V>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp); \
            else LK((k_loss_rows<L, 0, V>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp);                \
        } while (0)
        if (m->loss_long) G4R_LK_LOSS(true, 4);      // (rows that long always take four columns per thread)
        else { if (m->loss_quads) G4R_LK_LOSS(false, 4); else G4R_LK_LOSS(false, 1); }
#undef G4R_LK_LOSS
    }
    end();
    begin(KN_SCORE_BWD);
    if (lean_score_bwd(d)) {
        const LeanB& q = m->h_leanB;
        LK(k_score_b, dim3(q.nA + d.ksplit * q.nrb * q.ndb), dim3(512), 0, s, (const LeanB*)m->d_leanB, (const int*)(d.cur_in + 2 * B), (const int*)d.cur_col,
           (const float*)d.Sc, (const float*)d.hd[L - 1], (const float*)d.Wy, (float*)d.accWy, (unsigned)d.Dtop | ((unsigned)B << 16), (unsigned)d.N | ((unsigned)d.ldSc << 16));
    } else if (score_bmt_slabs(d, m->n_cu)) {
        const int ntile = d.ldSc / BMT_WA * (d.Dtop / 32);
        LK(k_score_bmt, dim3(2 * ntile), dim3(256), SMEM_BMT, s, (const float*)d.Sc, (const float*)d.hd[L - 1], (const float*)d.Wy, (const int*)d.col_item,
           (const int*)(d.cur_in + 2 * B), (const float*)d.zrow, dmp, (unsigned)d.Dtop | ((unsigned)(d.Dtop / 32) << 16), (unsigned)d.N | ((unsigned)d.ldSc << 16),
           (unsigned)B | ((unsigned)d.kch << 16), (unsigned)cdiv(B, 64) | ((unsigned)(d.Dtop / 128) << 16));
    } else if (score_bwd2(d)) {
        const int ndt = d.Dtop / 64, nrt = cdiv(B, 64);
        int nA = cdiv(d.ldSc, 64) * ndt, nB = d.ksplit * nrt * ndt, nC = cdiv(d.ldSc, 64);
        LK(k_score_bwd2, dim3(nA + nB + nC), dim3(GT_NTH), (size_t)(4 * 64 * 16) * sizeof(float) + (size_t)std::max(d.kch, 64) * sizeof(int), s, dmp, stp, nA, nB, ndt, nrt);
    } else if (wide_scores(d)) LK(k_score_bwd_w, dim3(m->nblkA + m->nblkB), dim3(GT_NTH), SMEM_SBW + (size_t)d.kch * sizeof(int), s, dmp, stp, m->nblkA, m->ndtA, m->ndtB, m->nrtB);
    else LK(k_score_bwd_n, dim3(m->nblkA + m->nblkB), dim3(GT_NTH), std::max(SMEM_TN, SMEM_NN) + (size_t)d.kch * sizeof(int), s, dmp, stp, m->nblkA, m->ndtA, m->ndtB, m->nrtB);
    end();
    for (int l = L - 1; l >= 0; --l) {
        if (lean_gru(d, l)) {
            const int nrb = cdiv(B, 16);
            begin(KN_GRU_DA);
            {
                const LeanDa& q = m->h_leanDa[l];
                LK(k_gru_da, dim3(cdiv(d.D[l], 16), nrb), dim3(128), 0, s, (const LeanDa*)(m->d_leanDa + l), (const int*)(d.cur_in + 2 * B), (const float*)q.dsrc,
                   (const float*)q.Wh, (const float*)q.z, (const float*)q.c, (const float*)q.H0, (const float*)q.H1, (unsigned)d.D[l] | ((unsigned)q.ks << 16), (unsigned)B);
            }
            end();
            begin(KN_GRU_DY);
            {
                const LeanDy& y = m->h_leanDy[l];
                LK(k_gru_dy, dim3(cdiv(d.IN[l], 16), nrb), dim3(1024), 0, s, (const LeanDy*)(m->d_leanDy + l), (const int*)(d.cur_in + 2 * B), (const int*)y.occ_idx,
                   (const float*)y.dV, (const float*)y.drp, (const float*)y.Wx, (const float*)y.r, (unsigned)d.D[l] | ((unsigned)d.IN[l] << 16), (unsigned)B);
            }
            end();
            continue;
        }
        if (fused_bwd(d, l)) {
            begin(KN_BWD_FUSED);
            LK(k_gru_bwd_fused, dim3(cdiv(d.IN[l], 32), cdiv(B, BF_ROWS)), dim3(512), smem_fused_bwd(d.D[l]), s, dmp, stp, l);
            end();
            continue;
        }
        begin(KN_BWD_PRE);
        LK(k_gru_bwd_pre, dim3(cdiv((long long)B * d.D[l], 256)), dim3(256), 0, s, dmp, stp, l);
        end();
        const g4r_model::WideGeo& G = m->wg[l];
        const int nrt64 = cdiv(B, 64);
        begin(KN_BWD_A);
        {
            dim3 ga(cdiv(d.D[l], GT_BN), cdiv(B, GT_BM));
            // behind k_score_bmt (raw gradient rows in the step plane) the top layer's launch carries their Adagrad rule: one extra
            // workgroup per 16 item rows (score_fin_rows)
            int nfin = 0;
            if (l == L - 1 && score_bmt_slabs(d, m->n_cu)) { nfin = cdiv(cdiv(d.N, 16), (int)ga.x); ga.y += nfin; }
            if (deep_geometry(m->ba_geo_env, m->n_cu, d.D[l], B)) LK(k_gru_bwd_a_w8d, ga, dim3(512), SMEM_BA_256, s, dmp, stp, l, nfin);
            else LK(k_gru_bwd_a_w4, ga, dim3(GT_NTH), SMEM_NT, s, dmp, stp, l, nfin);
        }
        end();
        begin(KN_BWD_B);
        if (l == 0 && d.embed_mode == G4R_EMBED_ONEHOT) LK(k_onehot_step, dim3(cdiv((long long)B * d.Ein, 4 * 256)), dim3(256), 0, s, dmp, stp);
        else if (G.use & 8) LK(k_gru_bwd_bw, dim3(cdiv(d.IN[l], 64) * nrt64 * G.bbn), dim3(256), SMEM_T3, s, dmp, stp, l, G.bbn, G.bbk);
        else LK(k_gru_bwd_b, dim3(cdiv(d.IN[l], GT_BN), cdiv(B, GT_BM)), dim3(GT_NTH_FEW), SMEM_BB, s, dmp, stp, l);
        end();
    }
    merged = merged_update(m) && !(recs && m->profile_split);      // g4r_profile(m, 2): the two roles of k_update as launches of their own
    if (merged) {
        // dense-gradient tiles (+ fused dense Adagrad on a single GPU; gradients to the RCCL buffer otherwise) and the sparse row
        // update in ONE launch (k_update): the two are independent, the all-reduce / dense apply of N > 1 follow behind
        const size_t smem = std::max(SMEM_TN, m->smem_sparse);
        const dim3 grid(m->ntiles + m->nblk_occ + 1), blk(SP_WAVES * 64);
        const bool one = row_chunks(d) == 1;
        if (d.bbn[0] > 0) {      // (dy of layer 0 as K-slice partial sums with the merged update: only when asked for, G4R_WIDE2)
            begin(KN_FINISH);
            LK(k_finish_rows, dim3(cdiv((long long)B * (d.IN[0] / 4), 256)), dim3(256), 0, s, dmp, stp);
            end();
        }
        begin(KN_UPDATE);
        const bool mo = d.mom > 0.f;
        if (lean_update(m) && !(d.bbn[0] > 0)) {
            const int nb8 = cdiv(d.R, 8);
            const unsigned packA = (unsigned)m->ntiles16 | ((unsigned)nb8 << 16), packB = (unsigned)d.R | ((unsigned)B << 16);
            const unsigned nbk = 1u + (unsigned)cdiv(d.ldSc, 512);
            const dim3 gl(nbk + m->ntiles16 + nb8);
            if (mo) LK(k_update_l<true>, gl, dim3(512), 0, s, (const LeanU*)m->d_leanU, (const DenseTile*)m->d_tiles16, (const int*)d.occ_idx, (int*)d.occ_fl,
                       (const float*)d.dSx, (const float*)d.dSy, (const float*)d.dSBy, packA, packB, nbk);
            else LK(k_update_l<false>, gl, dim3(512), 0, s, (const LeanU*)m->d_leanU, (const DenseTile*)m->d_tiles16, (const int*)d.occ_idx, (int*)d.occ_fl,
                    (const float*)d.dSx, (const float*)d.dSy, (const float*)d.dSBy, packA, packB, nbk);
            end();
            HIPCHK(hipGetLastError());
            return 0;
        }
#define G4R_LK_UPDATE(CH, DT_)                                                                                                          \
        do {                                                                                                                            \
            if (mo) LK((k_update<CH, DT_, true>), grid, blk, smem, s, dmp, stp, (const DenseTile*)m->d_tiles, m->ntiles, m->nblk_occ);  \
            else LK((k_update<CH, DT_, false>), grid, blk, smem, s, dmp, stp, (const DenseTile*)m->d_tiles, m->ntiles, m->nblk_occ);    \
        } while (0)
        if (one) G4R_LK_UPDATE(1, 32); else G4R_LK_UPDATE(2, 32);
#undef G4R_LK_UPDATE
        end();
        if (d.apply_dense_inplace || part == 1) { HIPCHK(hipGetLastError()); return 0; }
    } else {
    // (the dense-gradient tiles on a BRANCH of the step graph next to the sparse rows -- they share nothing -- were measured: the
    // fork / join costs more than the overlap gives, 126.6 -> 139.8 us per step at configs[2]; profiles/r05_experiments.md #8)
    begin(KN_DENSE);
    if (m->wide_dense) LK(k_dense_grad2, dim3(m->ntiles64 + (d.bbn[0] > 0 ? cdiv((long long)B * (d.IN[0] / 4), 256) : 0)), dim3(256), SMEM_T2K, s, dmp, stp, (const DenseTile*)m->d_tiles64, m->ntiles64);
    else {
        if (d.bbn[0] > 0) { LK(k_finish_rows, dim3(cdiv((long long)B * (d.IN[0] / 4), 256)), dim3(256), 0, s, dmp, stp); }
        LK(k_dense_grad<32>, dim3(m->ntiles), dim3(GT_NTH_FEW), SMEM_TN, s, dmp, stp, (const DenseTile*)m->d_tiles);
    }
    end();
    }
    }
    if (part == 1) { HIPCHK(hipGetLastError()); return 0; }
    if (part == 2) merged = merged_update(m) && !(recs && m->profile_split);
    // multi-rank: dense-gradient all-reduce, dense Adagrad, then the sparse embedding update, in stream order.
    // (running the first two on a stream of their own next to the sparse update -- which touches item rows only -- was measured on one
    // MI355X with a one-rank communicator: the two cross-stream event dependencies cost ~20 us per step, more than the ~11 us of sparse
    // update they can hide; that path was removed in round 6)
    const bool overlap = false;
    if (!d.apply_dense_inplace) {
        // staged dense path: (RCCL all-reduce when there are ranks) -> (global gradient norm -> clip factor, generic path with
        // grad_cap) -> dense rule on the flat gradient buffer
        const bool dist = !m->virtual_ranks && (m->cfg.nranks > 1 || m->comm_ready || m->p2p_ready);
        if (m->cfg.nranks > 1 && !m->comm_ready && !m->p2p_ready && !m->virtual_ranks) return fail("nranks > 1 but g4r_comm_init was not called");
        hipStream_t cs = overlap ? m->comm_stream : s;
        if (dist && !m->exact) {      // (exact-replica mode: the dense gradients travel with the all-gather of the occurrence blocks below)
            if (overlap) { HIPCHK(hipEventRecord(m->ev_fork, s)); HIPCHK(hipStreamWaitEvent(cs, m->ev_fork, 0)); }
            if (!overlap) { begin(KN_ALLREDUCE); if (recs) (void)hipEventRecord(cur_a, cs); }
            if (m->p2p_ready) hipLaunchKernelGGL(k_p2p_allreduce, dim3(m->p2p_nblk), dim3(256), 0, cs, m->p2p_args, (float*)d.dense_g);
            else NCCLCHK(ncclAllReduce(d.dense_g, d.dense_g, d.dense_count, ncclFloat, ncclSum, m->comm, cs));
            if (!overlap) { if (recs) (void)hipEventRecord(cur_b, cs); end(); }
        }
        if (d.generic && d.grad_cap > 0.f) {
            hipLaunchKernelGGL(k_grad_sqsum, dim3(G4R_NORM_BLOCKS), dim3(256), 0, cs, dmp, stp);
            hipLaunchKernelGGL(k_grad_clip, dim3(1), dim3(64), 0, cs, dmp);
        }
        // (generic optimizer path: the dense rule runs as extra workgroups of the sparse update's launch below)
        if (!d.generic) {
            if (!overlap) begin(KN_DENSE_APPLY);
            LK(k_dense_apply, dim3(cdiv(d.dense_count, 256)), dim3(256), 0, cs, (const DevModel*)m->d_dm);
            if (!overlap) end();
        }
        if (dist && overlap) HIPCHK(hipEventRecord(m->ev_join, cs));
    }
    if (d.generic) {
        // generic optimizer path: the sparse rule on raw per-occurrence gradients
        int nblk_g = m->nblk_occ_g;
        size_t smem_g = m->smem_sparse;
        if (m->exact) {
            // exact-replica mode: every rank's block of (occurrence list, gradient rows) to every rank, then the (last, first, count)
            // table of the concatenated list; the update below then runs over nranks * R occurrences, identically on every rank
            if (!m->virtual_ranks) {      // (virtual ranks: g4r_virtual_train_steps has copied the blocks)
                if (!m->comm_ready) return fail("sparse_exact needs the RCCL communicator (g4r_comm_init)");
                NCCLCHK(ncclAllGather((const float*)d.xbase + (size_t)m->cfg.rank * (size_t)d.xstride, (float*)d.xbase, (size_t)d.xstride, ncclFloat, m->comm, s));
            }
            const long long rlist = d.xmode == 3 ? (long long)d.xn * 2 * B + d.ns : (long long)d.R * d.xn;      // xlist_len
            hipLaunchKernelGGL(k_exact_occ, dim3(cdiv(rlist, 256)), dim3(256), 0, s, dmp);
            nblk_g = cdiv(rlist, SP_WAVES);
            smem_g = m->smem_exact;
        }
        begin(KN_SPARSE);
        const int nda = d.apply_dense_inplace ? 0 : cdiv(d.dense_count, SP_WAVES * 64);      // workgroups of the dense rule behind the row blocks
        if (row_chunks(d) == 1) LK(k_sparse_update_generic<1>, dim3(nblk_g + 1 + nda), dim3(SP_WAVES * 64), smem_g, s, dmp, stp, nblk_g, nda);
        else if (row_chunks(d) == 2) LK(k_sparse_update_generic<2>, dim3(nblk_g + 1 + nda), dim3(SP_WAVES * 64), smem_g, s, dmp, stp, nblk_g, nda);
        else LK(k_sparse_update_generic<4>, dim3(nblk_g + 1 + nda), dim3(SP_WAVES * 64), smem_g, s, dmp, stp, nblk_g, nda);
        end();
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (merged) { HIPCHK(hipGetLastError()); return 0; }
    begin(KN_SPARSE);
    {
        const bool mo = d.mom > 0.f;
        const dim3 grid(m->nblk_occ + 1), blk(SP_WAVES * 64);
#define G4R_LK_SPARSE(CH)                                                                                            \
        do {                                                                                                         \
            if (mo) LK((k_sparse_update<CH, true>), grid, blk, m->smem_sparse, s, dmp, stp, m->nblk_occ);            \
            else LK((k_sparse_update<CH, false>), grid, blk, m->smem_sparse, s, dmp, stp, m->nblk_occ);              \
        } while (0)
        if (row_chunks(d) == 1) G4R_LK_SPARSE(1); else if (row_chunks(d) == 2) G4R_LK_SPARSE(2); else G4R_LK_SPARSE(4);
#undef G4R_LK_SPARSE
    }
    end();
    if (overlap) HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0));
#undef begin
#undef LK
    HIPCHK(hipGetLastError());
    return 0;
}

static int apply_compaction(g4r_model* m, int64_t ci) {
    // gru4rec.py:647-651: H[i] <- H[i][valid_mask]; the current hidden state lives in H[l][gstep & 1]
    DevModel& d = m->dm;
    const int B = d.B;
    for (int l = 0; l < d.n_layers; ++l) {
        float* Hc = d.H[l][m->gstep & 1];
        const int W = d.D[l];
        hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((long long)B * W, 256)), dim3(256), 0, m->stream, m->d_tmpH, (const float*)Hc,
                           (const int*)(m->d_cmaps + ci * B), B, W);
        HIPCHK(hipMemcpyAsync(Hc, m->d_tmpH, (size_t)B * W * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    }
    return 0;
}

#define G4R_GRAPH_STEPS 16
#define G4R_GRAPH_STEPS_SMALL 4
// N > 1 (or the one-rank staged mode): the all-reduce is captured with the step, so that a replay covers 16 whole steps
// (kernels, RCCL all-reduce, dense apply) with no host work in between
// one GPU, staged dense path without a communicator (the generic optimizers: rmsprop / adadelta / adam / plain SGD / grad_cap): no
// collective in the step, so the whole step is captured like the fused single-GPU step (it used to replay a head graph and launch
// its tail eagerly; G4R_NO_LOCAL_GRAPH=1 keeps that)
static inline bool local_staged(const g4r_model* m) {
    return !m->dm.apply_dense_inplace && m->cfg.nranks <= 1 && !m->comm_ready && !m->p2p_ready && !m->virtual_ranks;
}
static inline bool dist_graph_wanted(const g4r_model* m) {
    return !m->dm.apply_dense_inplace && !m->dist_graph_failed &&
           (m->p2p_ready || m->comm_ready || local_staged(m));
}
static int ensure_graph(g4r_model* m) {
    if (m->gexec) return 0;
    const bool dist = !m->dm.apply_dense_inplace && !local_staged(m);
    const bool rccl_in_graph = dist && (!m->p2p_ready || (m->exact && m->comm_ready));      // (exact replicas: the step's collective is RCCL's all-gather even when the peer-memory all-reduce is attached)
    if (dist) {
        // RCCL sets its channels up on first use: that must not happen inside a capture (dense_g is scratch between steps)
        if (!m->p2p_ready) NCCLCHK(ncclAllReduce(m->dm.dense_g, m->dm.dense_g, m->dm.dense_count, ncclFloat, ncclSum, m->comm, m->stream));
        if (m->exact && m->comm_ready)      // the exact-replica step's collective is an all-gather: connect what THAT needs outside the capture, too
            NCCLCHK(ncclAllGather((const float*)m->dm.xbase + (size_t)m->cfg.rank * (size_t)m->dm.xstride, (float*)m->dm.xbase, (size_t)m->dm.xstride,
                                  ncclFloat, m->comm, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(m->stream, rccl_in_graph ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < G4R_GRAPH_STEPS && !rc; ++i) rc = launch_step(m, nullptr);
    hipError_t e = hipStreamEndCapture(m->stream, &graph);
    if (rc || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (!rc) fail(std::string("graph capture: ") + hipGetErrorString(e));
        return -1;
    }
    e = hipGraphInstantiate(&m->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { m->gexec = nullptr; (void)hipGetLastError(); return fail(std::string("graph instantiate: ") + hipGetErrorString(e)); }
    m->graph_steps = G4R_GRAPH_STEPS;
    if (!dist) {
        // a second, short graph: a run of 20 steps replays 16 + 4 instead of 16 + four eager steps (six launches each).  Best
        // effort: without it the remainder is launched eagerly as before.
        hipGraph_t g2 = nullptr;
        if (hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc2 = 0;
            for (int i = 0; i < G4R_GRAPH_STEPS_SMALL && !rc2; ++i) rc2 = launch_step(m, nullptr);
            hipError_t e2 = hipStreamEndCapture(m->stream, &g2);
            if (!rc2 && e2 == hipSuccess && g2 && hipGraphInstantiate(&m->gexec_small, g2, nullptr, nullptr, 0) != hipSuccess) m->gexec_small = nullptr;
            if (g2) (void)hipGraphDestroy(g2);
            (void)hipGetLastError();
        }
    }
    return 0;
}
// the step graph for this model: the whole step (single GPU; N > 1 with RCCL captured), or -- if RCCL cannot be captured on this
// runtime -- the head graph with an eager tail.  Returns 0 / -1; *whole tells which one is ready.
static int ensure_head_graph(g4r_model* m);
static int ensure_step_graph(g4r_model* m, bool* whole) {
    if (m->dm.apply_dense_inplace) { *whole = true; return ensure_graph(m); }
    if (dist_graph_wanted(m)) {
        if (ensure_graph(m) == 0) { *whole = true; return 0; }
        m->dist_graph_failed = true;
        fprintf(stderr, "[g4r] RCCL all-reduce could not be captured into the step graph (%s); launching it eagerly\n", g_err.c_str());
    }
    *whole = false;
    return ensure_head_graph(m);
}

static int ensure_head_graph(g4r_model* m) {
    if (m->gexec_head) return 0;
    hipGraph_t graph;
    HIPCHK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    if (launch_step(m, nullptr, 1)) { hipGraph_t g2; (void)hipStreamEndCapture(m->stream, &g2); return -1; }
    HIPCHK(hipStreamEndCapture(m->stream, &graph));
    HIPCHK(hipGraphInstantiate(&m->gexec_head, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    return 0;
}

int g4r_train_steps(g4r_model* m, int64_t t0, int64_t n_steps) {
    if (!m) return fail("null model");
    if (!m->d_in) return fail("no plan uploaded");
    if (t0 < 0 || n_steps < 0 || t0 + n_steps > m->T) return fail("step range outside the plan");
    if (m->dm.ns > 0 && !m->have_pop && !m->store_frozen) return fail("negative sampling needs g4r_set_popularity first");
    if (m->defer_on) {
        // k_defer_scan keys its newest-use table by the low 32 bits of the global step (signed atomicMax): refuse before they wrap
        // (27 h of training at 22 K steps/s on one handle; g4r_set_step_counters rebases the step and clears the table)
        if (m->gstep + n_steps >= ((int64_t)1 << 31) - 64) return fail("deferred row updates: the global step would pass 2^31 -- rebase it with g4r_set_step_counters (epoch boundary) or create the model without defer_updates");
        if (m->defer_broken) return fail("deferred row updates: an earlier call failed between a window's scan and its flush launch; the row updates pending then are lost -- recreate the model");
    }
    struct WindowGuard { g4r_model* m; bool open = false; ~WindowGuard() { if (open) m->defer_broken = true; } } wguard{m};
    HIPCHK(hipSetDevice(m->cfg.device));
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(512), 0, m->stream, (const DevModel*)m->d_dm, (StepState*)m->dm.st, (long long)t0, (long long)m->gstep);
    bool use_graph = m->cfg.use_graph && !m->profiling && !getenv("G4R_TRACE") && (m->dm.apply_dense_inplace || dist_graph_wanted(m));
    if (use_graph && !m->dm.apply_dense_inplace) {
        bool whole = false;
        if (ensure_step_graph(m, &whole)) return -1;
        use_graph = whole;
    }
    size_t ci = std::lower_bound(m->compact_steps.begin(), m->compact_steps.end(), t0) - m->compact_steps.begin();
    int64_t t = t0;
    const int64_t tend = t0 + n_steps;
    std::vector<EvRec> recs;
    while (t < tend) {
        // host-scheduled events that sit between steps: batch compaction, sample-store refill
        while (ci < m->compact_steps.size() && m->compact_steps[ci] == t) { if (apply_compaction(m, (int64_t)ci)) return -1; ++ci; }
        if (m->dm.ns > 0 && !m->store_frozen && m->gstep > 0 && m->gstep % m->gl == 0)
        {
            if (refill_store(m)) return -1;      // gru4rec.py:618-620
            hipLaunchKernelGGL(k_restage_inputs, dim3(1), dim3(512), 0, m->stream, (const DevModel*)m->d_dm, (StepState*)m->dm.st);
        }
        const bool devsync = m->sync_every_dev > 0 && m->comm_ready;
        if (devsync && m->since_sync >= m->sync_every_dev) {
            if (sync_dense_enqueue(m)) return -1;
            ++m->n_dev_syncs;
        }
        // steps until the next event
        int64_t run = tend - t;
        if (devsync) run = std::min<int64_t>(run, m->sync_every_dev - m->since_sync);
        if (ci < m->compact_steps.size()) run = std::min(run, m->compact_steps[ci] - t);
        if (m->dm.ns > 0 && !m->store_frozen) run = std::min<int64_t>(run, m->gl - (m->gstep % m->gl));
        if (run <= 0) return fail("internal: empty run");
        int64_t done = 0;
        // a deferral window around `nw` steps starting `done` steps into this run: which rows may wait (scan), ... steps ..., their flush
        auto window_open = [&](int64_t nw) {
            if (!m->defer_on) return;
            wguard.open = true;
            const dim3 gs(cdiv(nw * m->dm.R, 256));
            if (m->profiling) (void)hipEventRecord(m->ev_df[0], m->stream);
            hipLaunchKernelGGL(k_defer_scan, gs, dim3(256), 0, m->stream, (const DevModel*)m->d_dm, (long long)(t + done), (long long)(m->gstep + done), (int)nw, 0);
            hipLaunchKernelGGL(k_defer_scan, gs, dim3(256), 0, m->stream, (const DevModel*)m->d_dm, (long long)(t + done), (long long)(m->gstep + done), (int)nw, 1);
            if (m->profiling) (void)hipEventRecord(m->ev_df[1], m->stream);
        };
        auto window_close = [&](int64_t nw, int64_t first) -> int {
            if (!m->defer_on) return 0;
            if (m->profiling) (void)hipEventRecord(m->ev_df[2], m->stream);
            hipLaunchKernelGGL(k_sparse_flush, dim3(cdiv(nw * m->dm.dRcap, SP_WAVES * FL_NR)), dim3(SP_WAVES * 64), 0, m->stream, (const DevModel*)m->d_dm, (long long)(m->gstep + first), (int)nw);
            if (m->profiling) {
                (void)hipEventRecord(m->ev_df[3], m->stream);
                HIPCHK(hipStreamSynchronize(m->stream));
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, m->ev_df[0], m->ev_df[1]) == hipSuccess) { m->kn_ms[KN_SCAN] += ms; m->kn_n[KN_SCAN]++; }
                if (hipEventElapsedTime(&ms, m->ev_df[2], m->ev_df[3]) == hipSuccess) { m->kn_ms[KN_FLUSH] += ms; m->kn_n[KN_FLUSH]++; }
            }
            wguard.open = false;
            return 0;
        };
        if (use_graph && run >= G4R_GRAPH_STEPS_SMALL) {
            if (ensure_graph(m)) return -1;
            for (; done + m->graph_steps <= run; done += m->graph_steps) {
                window_open(m->graph_steps);
                HIPCHK(hipGraphLaunch(m->gexec, m->stream));
                if (window_close(m->graph_steps, done)) return -1;
            }
            if (m->gexec_small)
                for (; done + G4R_GRAPH_STEPS_SMALL <= run; done += G4R_GRAPH_STEPS_SMALL) {
                    window_open(G4R_GRAPH_STEPS_SMALL);
                    HIPCHK(hipGraphLaunch(m->gexec_small, m->stream));
                    if (window_close(G4R_GRAPH_STEPS_SMALL, done)) return -1;
                }
        }
        int64_t win_first = -1, win_n = 0;      // eager steps (no graph; per-kernel profiling): windows of up to G4R_DEFER_SLOTS steps
        for (; done < run; ++done) {
            if (m->defer_on && win_n == 0) {
                win_n = std::min<int64_t>(G4R_DEFER_SLOTS, run - done); win_first = done;
                window_open(win_n);
            }
            if (m->profiling) {
                // per-kernel durations: start/stop events attached to every dispatch (hipExtLaunchKernelGGL), i.e. the
                // kernel's own begin/end timestamps -- the quantity rocprofv3 --kernel-trace reports; eager launches
                recs.clear();
                if (launch_step(m, &recs)) return -1;
                HIPCHK(hipStreamSynchronize(m->stream));
                for (auto& r : recs) {
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { m->kn_ms[r.kn] += ms; m->kn_n[r.kn]++; }
                }
            } else if (m->cfg.use_graph && !m->dm.apply_dense_inplace && !getenv("G4R_TRACE")) {
                // N > 1: the step's compute kernels replay from a graph; the RCCL all-reduce, the dense apply and the
                // sparse update (two streams, fork/join events) are launched eagerly behind it
                if (ensure_head_graph(m)) return -1;
                HIPCHK(hipGraphLaunch(m->gexec_head, m->stream));
                if (launch_step(m, nullptr, 2)) return -1;
            } else if (launch_step(m, nullptr)) return -1;
            if (win_n > 0 && done + 1 == win_first + win_n) {
                if (window_close(win_n, win_first)) return -1;
                win_n = 0;
            }
        }
        t += run;
        m->gstep += run;
        m->since_sync += run;
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->p2p_ready) {
        unsigned late = 0;
        HIPCHK(hipMemcpy(&late, m->p2p_round + m->p2p_nblk, sizeof(late), hipMemcpyDeviceToHost));
        if (late) return fail("p2p all-reduce: a peer did not publish its gradients within G4R_P2P_TIMEOUT_MS (dead rank?)");
    }
    return 0;
}

// ---- virtual ranks ------------------------------------------------------------------------------------------------------
// n handles on ONE device stand in for the n ranks of a data-parallel run (each created with nranks = n, its own rank, its own
// plan): every step runs each handle's kernels up to the dense gradients, sums the n gradient buffers in rank order -- what the
// RCCL all-reduce delivers -- and lets each handle apply the sum (k_dense_apply divides by nranks) next to its GPU-local sparse
// update.  Item tables are reconciled by the caller with g4r_sync_export / g4r_sync_import.  Validation only (three stream
// synchronisations per step): the numbers it produces are what an n-GPU run computes, not how fast.
int g4r_virtual_train_steps(g4r_model* const* ms, int32_t n, int64_t t0, int64_t n_steps) {
    if (!ms || n < 1 || n > 16) return fail("virtual ranks: 1..16 handles");
    for (int q = 0; q < n; ++q) {
        g4r_model* m = ms[q];
        if (!m || !m->d_in) return fail("virtual ranks: null model / no plan uploaded");
        if (m->cfg.nranks != n || m->cfg.rank != q) return fail("virtual ranks: handle q must be created with rank = q, nranks = n");
        if (m->cfg.device != ms[0]->cfg.device || m->dm.dense_count != ms[0]->dm.dense_count || m->exact != ms[0]->exact || m->dm.xstride != ms[0]->dm.xstride)
            return fail("virtual ranks: handles differ");
        if (m->comm_ready || m->p2p_ready) return fail("virtual ranks: the handle already has a communicator / peer mappings");
        if (t0 < 0 || n_steps < 0 || t0 + n_steps > m->T) return fail("step range outside the plan");
        if (m->dm.ns > 0 && !m->have_pop && !m->store_frozen) return fail("negative sampling needs g4r_set_popularity first");
        m->virtual_ranks = true;
    }
    HIPCHK(hipSetDevice(ms[0]->cfg.device));
    g4r_model* m0 = ms[0];
    const int cnt = m0->dm.dense_count;
    if (!m0->d_vsum && dalloc(m0, &m0->d_vsum, (size_t)cnt)) return -1;
    VSumArgs va;
    memset(&va, 0, sizeof(va));
    std::vector<size_t> ci(n);
    for (int q = 0; q < n; ++q) {
        g4r_model* m = ms[q];
        va.src[q] = m->dm.dense_g; va.dst[q] = m->dm.dense_g;
        hipLaunchKernelGGL(k_set_state, dim3(1), dim3(512), 0, m->stream, (const DevModel*)m->d_dm, (StepState*)m->dm.st, (long long)t0, (long long)m->gstep);
        ci[q] = std::lower_bound(m->compact_steps.begin(), m->compact_steps.end(), t0) - m->compact_steps.begin();
    }
    for (int64_t t = t0; t < t0 + n_steps; ++t) {
        for (int q = 0; q < n; ++q) {
            g4r_model* m = ms[q];
            while (ci[q] < m->compact_steps.size() && m->compact_steps[ci[q]] == t) { if (apply_compaction(m, (int64_t)ci[q])) return -1; ++ci[q]; }
            if (m->dm.ns > 0 && !m->store_frozen && m->gstep > 0 && m->gstep % m->gl == 0) {
                if (refill_store(m)) return -1;
                hipLaunchKernelGGL(k_restage_inputs, dim3(1), dim3(512), 0, m->stream, (const DevModel*)m->d_dm, (StepState*)m->dm.st);
            }
            if (launch_step(m, nullptr, 1)) return -1;
        }
        for (int q = 0; q < n; ++q) HIPCHK(hipStreamSynchronize(ms[q]->stream));
        if (m0->exact) {
            // what the all-gather of the exact-replica mode delivers: every handle's own block into every other handle's buffer
            for (int q = 0; q < n; ++q)
                for (int p = 0; p < n; ++p)
                    if (p != q) HIPCHK(hipMemcpyAsync((float*)ms[q]->dm.xbase + (size_t)p * (size_t)m0->dm.xstride,
                                                      (const float*)ms[p]->dm.xbase + (size_t)p * (size_t)m0->dm.xstride,
                                                      (size_t)m0->dm.xstride * sizeof(float), hipMemcpyDeviceToDevice, ms[q]->stream));
            // (the next step's kernels of handle p rewrite p's block: every copy out of it must have run before p's tail is queued)
            for (int q = 0; q < n; ++q) HIPCHK(hipStreamSynchronize(ms[q]->stream));
        }
        if (!m0->exact) {      // (exact-replica mode: the blocks carry the dense gradients, every handle sums them itself)
            hipLaunchKernelGGL(k_virtual_sum, dim3(cdiv(cnt, 256)), dim3(256), 0, m0->stream, va, n, cnt, m0->d_vsum);
            hipLaunchKernelGGL(k_virtual_bcast, dim3(cdiv(cnt, 256)), dim3(256), 0, m0->stream, va, n, cnt, (const float*)m0->d_vsum);
            HIPCHK(hipStreamSynchronize(m0->stream));
        }
        for (int q = 0; q < n; ++q) {
            if (launch_step(ms[q], nullptr, 2)) return -1;
            ms[q]->gstep += 1;
        }
    }
    for (int q = 0; q < n; ++q) HIPCHK(hipStreamSynchronize(ms[q]->stream));
    return 0;
}

int g4r_get_losses(g4r_model* m, int64_t t0, int64_t n, float* out) {
    if (!m || !out) return fail("null argument");
    if (t0 < 0 || n < 0 || t0 + n > m->T) return fail("range outside the plan");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(out, m->d_loss + t0, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int g4r_synchronize(g4r_model* m) {
    if (!m) return fail("null model");
    HIPCHK(hipSetDevice(m->cfg.device));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
int64_t g4r_global_step(g4r_model* m) { return m ? m->gstep : -1; }
int64_t g4r_refills(g4r_model* m) { return m ? (int64_t)m->refills : -1; }
// resume: continue the counter-based random streams (dropout masks are keyed by the global step, the sample store by its refill
// number) where a checkpointed run stopped; the store is regenerated as that run's last refill left it
int g4r_set_step_counters(g4r_model* m, int64_t global_step, int64_t refills) {
    if (!m) return fail("null model");
    if (global_step < 0 || refills < 0) return fail("negative counter");
    HIPCHK(hipSetDevice(m->cfg.device));
    m->gstep = global_step;
    if (m->defer_on)      // (the scan's "newest step that gathers the item" table is keyed by the global step)
        HIPCHK(hipMemsetAsync(m->dm.last_use, 0, (size_t)(m->cfg.embed_mode != G4R_EMBED_CONSTRAINED ? 2 : 1) * m->dm.n_items * sizeof(int), m->stream));
    if (m->dm.ns > 0 && !m->store_frozen) {
        if (!m->have_pop) return fail("g4r_set_popularity first");
        if (refills < 1) return fail("a model with negative sampling has filled its store at least once");
        m->refills = (unsigned)(refills - 1);
        if (refill_store(m)) return -1;
        HIPCHK(hipStreamSynchronize(m->stream));
    } else {
        m->refills = (unsigned)refills;
    }
    return 0;
}
int g4r_profile(g4r_model* m, int32_t enable) {
    if (!m) return fail("null model");
    m->profiling = enable != 0;
    m->profile_split = enable == 2;      // the sparse row update timed ALONE (k_sparse_update next to k_dense_grad instead of the merged k_update)
    if (enable) for (int i = 0; i < KN_COUNT; ++i) { m->kn_ms[i] = 0; m->kn_n[i] = 0; }
    return 0;
}
int g4r_kernel_time(g4r_model* m, int32_t which, const char** name, double* total_ms, int64_t* launches) {
    if (!m || which < 0 || which >= KN_COUNT) return fail("bad kernel index");
    if (name) *name = KN_NAMES[which];
    if (total_ms) *total_ms = m->kn_ms[which];
    if (launches) *launches = m->kn_n[which];
    return 0;
}

int g4r_reset_hidden(g4r_model* m) {
    if (!m) return fail("null model");
    HIPCHK(hipSetDevice(m->cfg.device));
    for (int l = 0; l < m->dm.n_layers; ++l)
        for (int q = 0; q < 2; ++q)
            HIPCHK(hipMemsetAsync(m->dm.H[l][q], 0, (size_t)m->dm.B * m->dm.D[l] * sizeof(float), m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
