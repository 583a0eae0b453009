// g4r_host_create.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: g4r_device_count / g4r_sizeof_config / g4r_create (the memory plan: every buffer of the step is allocated here) / g4r_destroy.

int g4r_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char* g4r_last_error(void) { return g_err.c_str(); }
#ifndef G4R_HIPCC_VERSION
#define G4R_HIPCC_VERSION "unknown"
#endif
// library version, target, and the hipcc the device code was generated with (gru4rec_amd/build.py passes it; the same build
// audits the generated code for premature uses of hand-counted asm loads and refuses to install a library that has one)
const char* g4r_version(void) { return "gru4rec_hip 0.4 (gfx950; hipcc " G4R_HIPCC_VERSION "; isa-audited)"; }
int g4r_sizeof_config(void) { return (int)sizeof(g4r_config); }

// argument blocks of the narrow-layer kernels (g4r_lean_kernels.cuh): everything they read from the model, from pointers that are final here
static int build_lean_args(g4r_model* m) {
    DevModel& d = m->dm;
    const int L = d.n_layers;
    std::vector<LeanV> av(L); std::vector<LeanH> ah(L); std::vector<LeanDa> aa(L); std::vector<LeanDy> ay(L);
    bool any = false;
    const bool constrained_all = d.embed_mode == G4R_EMBED_CONSTRAINED;
    for (int l = 0; l < L; ++l) {
        if (!lean_gru(d, l)) continue;
        any = true;
        const bool constrained = d.embed_mode == G4R_EMBED_CONSTRAINED;
        LeanV& v = av[l]; memset(&v, 0, sizeof(v));
        v.Wx = d.dense_p + d.offWx[l]; v.Wrz = d.dense_p + d.offWrz[l]; v.Bh = d.dense_p + d.offBh[l];
        v.H0 = d.H[l][0]; v.H1 = d.H[l][1];
        v.ysrc = (l == 0) ? (constrained ? d.Wy : d.E) : d.hd[l - 1];
        v.cur_in = d.cur_in; v.Vc = d.Vc[l]; v.r = d.r[l]; v.Hr = d.Hr[l]; v.z = d.z[l]; v.yin0 = d.yin0;
        v.occ_idx = d.occ_idx; v.occ_fl = d.occ_fl + 4 * (constrained ? (size_t)0 : (size_t)d.n_items);
        v.st = d.st; v.seed = d.seed; v.B = d.B; v.D = d.D[l]; v.IN = d.IN[l]; v.R = d.R; v.first = (l == 0) ? 1 : 0; v.pub_fl = d.xmode == 0 ? 1 : 0;
        v.drop_e = d.drop_e; v.dbg = d.dbgclk; v.dbgtile = d.dbgtile; v.n_items = d.n_items;
        LeanH& h = ah[l]; memset(&h, 0, sizeof(h));
        h.Wh = d.dense_p + d.offWh[l]; h.H0 = d.H[l][0]; h.H1 = d.H[l][1]; h.Hr = d.Hr[l]; h.Vc = d.Vc[l]; h.z = d.z[l];
        h.cur_rst = d.cur_in + d.B; h.c = d.c[l]; h.hd = d.hd[l]; h.st = d.st; h.seed = d.seed; h.B = d.B; h.D = d.D[l];
        h.hidden_act = d.hidden_act; h.stream = (int)(G4R_STREAM_DROP_HIDDEN + (unsigned)l); h.ha_p0 = d.ha_p0; h.ha_p1 = d.ha_p1; h.drop_h = d.drop_h; h.dbg = d.dbgclk; h.dbgtile = d.dbgtile;
        LeanDa& q = aa[l]; memset(&q, 0, sizeof(q));
        q.Wh = h.Wh; q.H0 = h.H0; q.H1 = h.H1; q.z = d.z[l]; q.c = d.c[l];
        if (l == L - 1) { q.dsrc = d.dhpart; q.ks = d.ksplit; }
        else if (d.bbn[l + 1] > 0) { q.dsrc = d.dyp; q.ks = d.bbn[l + 1]; }
        else { q.dsrc = d.dyl[l]; q.ks = 1; }
        q.dV = d.dV[l]; q.drp = d.drp; q.st = d.st; q.seed = d.seed; q.B = d.B; q.D = d.D[l]; q.hidden_act = d.hidden_act; q.stream = h.stream;
        q.ha_p0 = d.ha_p0; q.ha_p1 = d.ha_p1; q.drop_h = d.drop_h; q.dbg = d.dbgclk; q.dbgtile = d.dbgtile;
        LeanDy& y = ay[l]; memset(&y, 0, sizeof(y));
        y.Wx = v.Wx; y.H0 = h.H0; y.H1 = h.H1; y.r = d.r[l]; y.drp = d.drp; y.dV = d.dV[l]; y.occ_idx = d.occ_idx; y.occ_fl = v.occ_fl;
        y.accT = constrained ? d.accWy : d.accE; y.dSx = d.dSx; y.dAx = d.dAx; y.dylo = (l > 0) ? d.dyl[l - 1] : nullptr;
        y.st = d.st; y.seed = d.seed; y.dSx_stride = d.dSx_stride; y.B = d.B; y.D = d.D[l]; y.IN = d.IN[l]; y.layer0 = (l == 0) ? 1 : 0;
        y.generic = d.generic; y.defer_mask = d.defer_mask; y.lr = d.lr; y.drop_e = d.drop_e; y.dbg = d.dbgclk; y.dbgtile = d.dbgtile; y.n_items = d.n_items;
    }
    if (lean_scores(d)) {
        LeanS q; memset(&q, 0, sizeof(q));
        q.col_item = d.col_item; q.occ_idx = d.occ_idx + d.B; q.occ_fl = d.occ_fl; q.mp = nullptr; q.dbg = d.dbgclk; q.dbgtile = d.dbgtile; q.R = d.R; q.pub_fl = d.xmode == 0 ? 1 : 0; q.logq = d.logq;
        m->h_leanS = q;
        any = true;
    }
    if (lean_score_bwd(d)) {
        LeanB q; memset(&q, 0, sizeof(q));
        q.accBy = d.accBy; q.occ_fl = d.occ_fl; q.dSy = d.dSy; q.dAy = d.dAy; q.dSBy = d.dSBy; q.dABy = d.dABy; q.dhpart = d.dhpart; q.dbg = d.dbgclk; q.dbgtile = d.dbgtile;
        q.dSy_stride = d.dSy_stride; q.dSBy_stride = d.dSBy_stride; q.defer_mask = d.defer_mask; q.generic = d.generic;
        q.ndh = cdiv(d.Dtop + 1, 64); q.nA = cdiv(d.ldSc, 16) * q.ndh; q.nrb = cdiv(d.B, 16); q.ndb = cdiv(d.Dtop, 64); q.lr = d.lr;
        m->h_leanB = q;
    }
    if (d.apply_dense_inplace && d.B <= 128 && std::max(d.Dtop, d.Ein) <= 256 && !d.generic) {
        // k_update_l: argument block + its table of 16 x 64 dense tiles
        LeanU u; memset(&u, 0, sizeof(u));
        u.mp = nullptr; u.st = d.st; u.Wy = d.Wy; u.E = d.E; u.accWy = d.accWy; u.accE = d.accE; u.velWy = d.velWy; u.velE = d.velE;
        u.By = d.By; u.accBy = d.accBy; u.velBy = d.velBy; u.dAx = d.dAx; u.dAy = d.dAy; u.dABy = d.dABy;
        u.dense_p = d.dense_p; u.dense_acc = d.dense_acc; u.dense_vel = d.dense_vel; u.yin0 = d.yin0; u.meta = d.cur_in + 2 * d.B;
        u.dbg = d.dbgclk; u.dbgtile = d.dbgtile; u.n_items = d.n_items; u.constrained = constrained_all ? 1 : 0; u.wE = d.Ein; u.wY = d.Dtop;
        u.lr = d.lr; u.mom = d.mom; u.lmbd = d.lmbd;
        m->h_leanU = u;
        std::vector<DenseTile> tiles;
        for (int l = 0; l < L; ++l) {
            const int D = d.D[l], IN = d.IN[l];
            auto add = [&](const float* x0, const float* x1, int ldx, int nrows, int ncols, int coff, int ldo, long long base) {
                for (int r = 0; r < nrows; r += 16)
                    for (int c = 0; c < ncols; c += 64) {
                        DenseTile t;
                        t.X0 = x0; t.X1 = x1; t.dV = d.dV[l]; t.base = base; t.ldx = ldx; t.ldv = 3 * D; t.nrows = nrows;
                        t.ncols = ncols; t.coff = coff; t.ldo = ldo; t.r0 = r; t.c0 = c; t.gather = (x0 == nullptr && nrows > 1) ? 1 : 0; t.pad = 0;
                        tiles.push_back(t);
                    }
            };
            const float* yin = (l == 0) ? nullptr : d.hd[l - 1];
            if (!(l == 0 && d.embed_mode == G4R_EMBED_ONEHOT)) add(yin, yin, IN, IN, 3 * D, 0, 3 * D, d.offWx[l]);
            add(d.Hr[l], d.Hr[l], D, D, D, 0, D, d.offWh[l]);
            add(d.H[l][0], d.H[l][1], D, D, 2 * D, D, 2 * D, d.offWrz[l]);
            add(nullptr, nullptr, 0, 1, 3 * D, 0, 3 * D, d.offBh[l]);
        }
        m->ntiles16 = (int)tiles.size();
        if (dalloc(m, &m->d_tiles16, tiles.size()) || dalloc(m, &m->d_leanU, (size_t)1)) return -1;
        HIPCHK(hipMemcpyAsync(m->d_tiles16, tiles.data(), tiles.size() * sizeof(DenseTile), hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        any = true;
    }
    m->h_leanV = av; m->h_leanH = ah; m->h_leanDa = aa; m->h_leanDy = ay;      // (host copies: launch_step passes their hot fields as kernel arguments)
    if (!any) return 0;
    if (lean_score_bwd(d)) {
        if (dalloc(m, &m->d_leanB, (size_t)1)) return -1;
        HIPCHK(hipMemcpyAsync(m->d_leanB, &m->h_leanB, sizeof(LeanB), hipMemcpyHostToDevice, m->stream));
    }
    if (lean_scores(d)) {
        if (dalloc(m, &m->d_leanS, (size_t)1)) return -1;
        m->h_leanS.mp = nullptr;      // (set below: the descriptor is allocated after this function)
    }
    if (dalloc(m, &m->d_leanV, (size_t)L) || dalloc(m, &m->d_leanH, (size_t)L) || dalloc(m, &m->d_leanDa, (size_t)L) || dalloc(m, &m->d_leanDy, (size_t)L)) return -1;
    HIPCHK(hipMemcpyAsync(m->d_leanV, av.data(), L * sizeof(LeanV), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_leanH, ah.data(), L * sizeof(LeanH), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_leanDa, aa.data(), L * sizeof(LeanDa), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_leanDy, ay.data(), L * sizeof(LeanDy), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

int g4r_create(const g4r_config* cfg, g4r_model** out) {
    if (!cfg || !out) return fail("null argument");
    if (cfg->n_layers < 1 || cfg->n_layers > G4R_MAX_LAYERS) return fail("n_layers out of range");
    if (cfg->batch_size < 1 || cfg->n_items < 1) return fail("batch_size / n_items must be positive");
    for (int l = 0; l < cfg->n_layers; ++l)
        if (cfg->layers[l] % 4 != 0 || cfg->layers[l] < 4 || cfg->layers[l] > 1024)
            return fail("layer sizes must be multiples of 4 in [4, 1024]");
    if (cfg->embed_mode == G4R_EMBED_SEPARATE && (cfg->embedding % 4 != 0 || cfg->embedding < 4 || cfg->embedding > 1024))
        return fail("embedding must be a multiple of 4 in [4, 1024]");
    if (cfg->embed_mode != G4R_EMBED_CONSTRAINED && cfg->embed_mode != G4R_EMBED_SEPARATE && cfg->embed_mode != G4R_EMBED_ONEHOT)
        return fail("unsupported embedding mode");
    if (cfg->embed_mode == G4R_EMBED_ONEHOT && 3 * cfg->layers[0] > 1024)
        return fail("one-hot input: 3 * layers[0] must be <= 1024 (row width of the Wx[0] table)");
    if (cfg->loss < 0 || cfg->loss > G4R_LOSS_XE_LOGIT) return fail("unsupported loss");
    if (cfg->smoothing != 0.f && cfg->loss != G4R_LOSS_XE && cfg->loss != G4R_LOSS_XE_LOGIT) return fail("smoothing needs a cross-entropy loss");
    if (cfg->hidden_act == G4R_ACT_SOFTMAX_LOGIT) return fail("softmax_logit is not a hidden activation");
    if (cfg->adapt < 0 || cfg->adapt > G4R_ADAPT_NONE) return fail("unknown adapt");
    if (cfg->grad_cap < 0.f) return fail("grad_cap must be >= 0");
    if (cfg->hidden_act == G4R_ACT_SOFTMAX) return fail("softmax is not a hidden activation");
    int ndev = g4r_device_count();
    if (ndev <= 0) return fail("no HIP device visible: the gfx950 path has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));
    int n_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device));
    g4r_model* m = new g4r_model();
    m->cfg = *cfg;
    m->n_cu = std::max(n_cu, 1);
    m->p2_geo_env = env_int("G4R_P2_GEO", -1);
    m->ba_geo_env = env_int("G4R_BA_GEO", -1);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return fail("stream create"); }
    if (hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess) { g4r_destroy(m); return fail("stream create"); }
    DevModel& d = m->dm;
    memset(&d, 0, sizeof(d));
    const int L = cfg->n_layers, B = cfg->batch_size;
    d.n_items = cfg->n_items; d.n_layers = L; d.B = B;
    // negatives: generate_length = sample_store // n_sample ; a store of <= 1 rows means "no store" (gru4rec.py:546-550), i.e. a
    // fresh row of negatives for every step (:614-615): a one-row store that is refilled before every step
    const int ns = std::max(cfg->n_sample, 0);
    int64_t gl = (ns > 0 && cfg->sample_store > 0) ? cfg->sample_store / ns : 0;
    if (ns > 0 && gl <= 1) gl = 1;
    m->gl = gl;
    d.ns = ns; d.N = B + ns; d.R = 2 * B + ns; d.ldSc = (d.N + 15) & ~15;
    d.gl = (int)std::max<int64_t>(gl, 1);
    d.loss = cfg->loss; d.final_act = cfg->final_act; d.hidden_act = cfg->hidden_act; d.embed_mode = cfg->embed_mode;
    d.fa_p0 = cfg->final_act_p0; d.fa_p1 = cfg->final_act_p1; d.ha_p0 = cfg->hidden_act_p0; d.ha_p1 = cfg->hidden_act_p1;
    d.lr = cfg->learning_rate; d.mom = cfg->momentum; d.lmbd = cfg->lmbd; d.bpreg = cfg->bpreg; d.logq = cfg->logq;
    d.inv_B = 1.0f / (float)B;
    d.smoothing = cfg->smoothing;
    d.adapt = cfg->adapt; d.ap0 = cfg->adapt_p0; d.ap1 = cfg->adapt_p1; d.grad_cap = cfg->grad_cap;
    // exact-replica mode of N > 1: raw per-occurrence gradients (the generic path's producers), exchanged every step
    // (G4R_FORCE_STAGED=1: the N > 1 data path with a one-rank communicator -- what a 1-GPU box can run and time of it)
    const bool exact = cfg->sparse_exact != 0 && (cfg->nranks > 1 || getenv("G4R_FORCE_STAGED") != nullptr);
    if (cfg->sparse_exact != 0 && cfg->grad_cap > 0.f) { g4r_destroy(m); return fail("sparse_exact does not support grad_cap (the norm would be per rank)"); }
    m->exact = exact;
    d.generic = (cfg->adapt != G4R_ADAPT_ADAGRAD || cfg->grad_cap > 0.f || exact) ? 1 : 0;
    d.drop_h = cfg->dropout_p_hidden; d.drop_e = cfg->dropout_p_embed;
    // dropout masks are keyed by (seed, step, row, column) with LOCAL rows: in exact-replica mode the ranks share cfg->seed (ONE stream of
    // negatives: refill_store), so the masks take a rank-specific key -- the nranks x B rows of the joint batch must not repeat one pattern
    d.seed = cfg->seed + ((cfg->sparse_exact != 0 && cfg->nranks > 1) ? 7919ull * (unsigned long long)cfg->rank : 0ull);
    d.Dtop = cfg->layers[L - 1];
    // width of the layer-0 input rows: shared Wy rows, E rows, or (one-hot input) rows of Wx[0] = [cand|r|z] pre-activations
    d.Ein = (cfg->embed_mode == G4R_EMBED_CONSTRAINED) ? d.Dtop : (cfg->embed_mode == G4R_EMBED_ONEHOT ? 3 * cfg->layers[0] : cfg->embedding);
    int off = 0;
    for (int l = 0; l < L; ++l) {
        d.D[l] = cfg->layers[l];
        d.IN[l] = (l == 0) ? (cfg->embed_mode == G4R_EMBED_ONEHOT ? 0 : d.Ein) : cfg->layers[l - 1];
        d.offWx[l] = off; off += d.IN[l] * 3 * d.D[l];
        d.offWh[l] = off; off += d.D[l] * d.D[l];
        d.offWrz[l] = off; off += d.D[l] * 2 * d.D[l];
        d.offBh[l] = off; off += 3 * d.D[l];
    }
    d.dense_count = off;
    // G4R_FORCE_STAGED=1: exercise the multi-rank data path (gradient staging -> RCCL -> k_dense_apply) on one GPU
    d.apply_dense_inplace = (cfg->nranks <= 1 && !getenv("G4R_FORCE_STAGED") && !d.generic) ? 1 : 0;
    d.grad_scale = 1.0f / (float)std::max(cfg->nranks, 1);
    const size_t I = cfg->n_items;
#define DA(p, n) if (dalloc(m, &(p), (n))) { g4r_destroy(m); return -1; }
    DA(d.dense_p, off); DA(d.dense_acc, off); DA(d.dense_vel, off); DA(d.dense_g, off);
    DA(d.Wy, I * d.Dtop); DA(d.accWy, I * d.Dtop); DA(d.By, I); DA(d.accBy, I);
    if (cfg->momentum > 0.f) { DA(d.velWy, I * d.Dtop); DA(d.velBy, I); }
    if (cfg->embed_mode != G4R_EMBED_CONSTRAINED) {     // E table, or Wx[0] as a row table (one-hot input)
        DA(d.E, I * d.Ein); DA(d.accE, I * d.Ein);
        if (cfg->momentum > 0.f) DA(d.velE, I * d.Ein);
    }
    if (d.generic) {
        const bool two = (cfg->adapt == G4R_ADAPT_ADADELTA || cfg->adapt == G4R_ADAPT_ADAM), cnt = (cfg->adapt == G4R_ADAPT_ADAM);
        if (two) { DA(d.acc2Wy, I * d.Dtop); DA(d.acc2By, I); DA(d.dense_acc2, off); if (d.E) DA(d.acc2E, I * d.Ein); }
        if (cnt) { DA(d.cntWy, I * d.Dtop); DA(d.cntBy, I); DA(d.dense_cnt, off); if (d.E) DA(d.cntE, I * d.Ein); }
        DA(d.gsq_part, G4R_NORM_BLOCKS); DA(d.gclip, 1);
        const float one = 1.f;
        if (hipMemcpyAsync(d.gclip, &one, sizeof(float), hipMemcpyHostToDevice, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) {
            g4r_destroy(m); return fail("gclip init");
        }
    }
    int maxD = 0;
    for (int l = 0; l < L; ++l) {
        const size_t bd = (size_t)B * d.D[l];
        maxD = std::max(maxD, d.D[l]);
        DA(d.H[l][0], bd); DA(d.H[l][1], bd);
        DA(d.r[l], bd); DA(d.z[l], bd); DA(d.c[l], bd); DA(d.hd[l], bd); DA(d.Hr[l], bd);
        DA(d.dV[l], bd * 3); DA(d.dyl[l], bd); DA(d.Vc[l], bd);
    }
    DA(m->d_tmpH, (size_t)B * maxD);
    {      // narrow layers: the K-slice partial planes of dr' (k_gru_da -> k_gru_dy)
        size_t drp_floats = 0;
        for (int l = 0; l < L; ++l) if (lean_gru(d, l)) drp_floats = std::max(drp_floats, (size_t)cdiv(d.D[l], 16) * B * d.D[l]);
        if (drp_floats) DA(d.drp, drp_floats);
    }
    DA(d.yin0, (size_t)B * std::max(d.IN[0], 4));
    DA(d.Sc, (size_t)B * d.ldSc);
    {
        // occ_idx | dSx | dSy | dSBy of this rank in ONE block (DevModel::xbase): what the exact-replica mode all-gathers every step.
        // Offsets are multiples of 64 floats (16-byte rows stay aligned); occ_idx is staged with 16-byte loads up to Rpad.
        auto up64 = [](size_t n) { return (n + 63) & ~(size_t)63; };
        const size_t nOcc = up64((size_t)((d.R + 255) & ~255) + 256 + 64);
        d.xoffSx = (int)nOcc;
        d.xoffSy = (int)(nOcc + up64((size_t)B * d.Ein));
        d.xoffSBy = (int)(d.xoffSy + up64((size_t)d.ldSc * d.Dtop));
        // exact-replica mode: the rank's raw dense gradients ride in the same block (ONE collective per step: the all-gather
        // replaces the all-reduce, every rank adds the ranks' gradients up itself, in rank order -- dense_apply_elem)
        d.xoffDg = (int)(d.xoffSBy + up64((size_t)d.ldSc));
        d.xstride = (long long)(d.xoffDg + (exact ? up64((size_t)d.dense_count) : 0));
        d.xn = exact ? cfg->nranks : 1;
        d.xmode = exact ? std::min(std::max(cfg->sparse_exact, 1), 3) : 0;
        float* xb = nullptr;
        DA(xb, (size_t)d.xn * (size_t)d.xstride);
        d.xbase = xb;
        float* own = xb + (size_t)(exact ? cfg->rank : 0) * (size_t)d.xstride;
        d.occ_idx = (int*)own; d.dSx = own + d.xoffSx; d.dSy = own + d.xoffSy; d.dSBy = own + d.xoffSBy;
        if (exact) d.dense_g = own + d.xoffDg;      // (the buffer allocated above stays unused)
    }
    DA(d.dAx, (size_t)B * d.Ein); DA(d.dAy, (size_t)d.ldSc * d.Dtop); DA(d.dABy, d.ldSc);
    // Deferred row updates (g4r_step_kernels.cuh: k_defer_scan / k_sparse_flush): the single-GPU Adagrad step without momentum / L2 term, replayed
    // from the step graph.  The step planes become rings of G4R_GRAPH_STEPS slots (one window = one graph replay).  OPT-IN (G4R_DEFER=1;
    // GRU4Rec.defer_updates, bench.py --defer): bit-identical results and a flush launch at 59 % of the HBM peak on the bytes it
    // moves at BASELINE configs[2] -- but the step gets 2-5 % SLOWER, because the update launch it relieves is at its latency floor
    // (cfg3: k_sparse_update 7.5 -> 6.2 us with 90 % of the rows gone) or bound by its dense-gradient tiles (cfg4), and the flush
    // (2.9 / 7.4 us per step) and scan (0.7 / 1.1) come on top (profiles/r05_experiments.md #7).
    m->lean_upd = env_int("G4R_LEAN_UPDATE", 1) != 0;
    m->defer_on = d.apply_dense_inplace && !d.generic && cfg->momentum <= 0.f && cfg->lmbd == 0.f && env_int("G4R_DEFER", cfg->defer_updates) != 0;
    if (m->defer_on) {
        const size_t W = G4R_DEFER_SLOTS;
        d.defer_mask = (int)W - 1;
        d.dRcap = cdiv(d.R, SP_WAVES) * SP_WAVES;
        d.dSx_stride = (long long)(((size_t)B * d.Ein + 63) & ~(size_t)63);
        d.dSy_stride = (long long)(((size_t)d.ldSc * d.Dtop + 63) & ~(size_t)63);
        d.dSBy_stride = (long long)(((size_t)d.ldSc + 63) & ~(size_t)63);
        float *rx = nullptr, *ry = nullptr, *rb = nullptr;
        DA(rx, W * (size_t)d.dSx_stride); DA(ry, W * (size_t)d.dSy_stride); DA(rb, W * (size_t)d.dSBy_stride);
        d.dSx = rx; d.dSy = ry; d.dSBy = rb;
        DA(d.last_use, (size_t)(cfg->embed_mode != G4R_EMBED_CONSTRAINED ? 2 : 1) * I);
        DA(d.dcand, W * (size_t)d.dRcap); DA(d.dlist, W * (size_t)d.dRcap); DA(d.dstat, 2048);
        if (hipMemsetAsync(d.dlist, 0xFF, W * (size_t)d.dRcap * sizeof(int), m->stream) != hipSuccess) { g4r_destroy(m); return fail("dlist init"); }
        for (auto& e : m->ev_df) if (hipEventCreate(&e) != hipSuccess) { g4r_destroy(m); return fail("event create"); }
    }
    DA(d.lossrow, B);
    DA(d.col_item, d.ldSc); DA(d.cur_in, 2 * (size_t)B + 8); DA(d.cur_col, d.ldSc);
    DA(d.occ_fl, (size_t)(cfg->embed_mode != G4R_EMBED_CONSTRAINED ? 2 : 1) * I * 4);
    DA(d.st, 1);
    // scoring backward geometry: role A tiles (n x d, one spare d column for dSBy), role B tiles (b x d x k-chunk)
    {
        // k_gru_bwd_fused sums the slabs next to everything else it loads: half as many, twice as deep (k_score_bwd +0.4 us at cfg2)
        const int slabs_target = (fused_bwd(d, d.n_layers - 1) || lean_gru(d, d.n_layers - 1)) ? 9 : 17;
        d.kch = GT_BK * std::max(1, (cdiv(d.ldSc, GT_BK) + slabs_target / 2) / slabs_target);      // ~17 slabs whatever the number of negatives
        if (score_bwd2(d)) {
            // k_score_bwd2: its 64 x 64 tiles cost microseconds of MFMA each and all of them are resident at once, so the launch
            // lasts as long as the CU with one tile more than the others.  The number of dh slabs is free: take the one (12..24)
            // that makes role A + role B tiles fill whole rounds of CUs best (B = 512, N = 8704, D = 256: 17 slabs = 1088 tiles
            // 64.2 us, 15 slabs = 1024 tiles 60.6 us).  Slab depth only needs the 16-byte alignment of the row loads.
            const int ndt = d.Dtop / 64, nrt = cdiv(B, 64), nA = cdiv(d.ldSc, 64) * ndt;
            double best = 2.0;
            for (int ks = 12; ks <= 24; ++ks) {
                const int kch = (cdiv(d.ldSc, ks) + 7) & ~7;
                if (cdiv(d.ldSc, kch) != ks) continue;
                const double rounds = (double)(nA + ks * nrt * ndt) / m->n_cu;
                const double waste = (std::ceil(rounds) - rounds) / std::ceil(rounds) + 1e-3 * std::abs(ks - 17);
                if (waste < best) { best = waste; d.kch = kch; }
            }
        }
        if (score_bmt_slabs(d, m->n_cu)) d.kch = d.ldSc / score_bmt_slabs(d, m->n_cu);      // k_score_bmt: as many role-B as role-A tiles
        if (lean_score_bwd(d)) d.kch = 128;      // k_score_b: slabs of 128 score columns (eight waves x 16)
        d.ksplit = cdiv(d.ldSc, d.kch);
        DA(d.dhpart, (size_t)d.ksplit * B * d.Dtop);
        const int TB = wide_scores(d) ? 64 : 32;      // tile edge of k_score_bwd
        m->ndtA = cdiv(d.Dtop + 1, TB);
        m->nblkA = cdiv(d.ldSc, TB) * m->ndtA;
        m->ndtB = cdiv(d.Dtop, TB);
        m->nrtB = cdiv(B, TB);
        m->nblkB = d.ksplit * m->nrtB * m->ndtB;
        m->nblk_occ = cdiv(d.R, SP_WAVES);
        m->nblk_occ_g = m->nblk_occ;      // generic optimizer path (one occurrence per wave; exact-replica mode: sized at launch)
        m->smem_sparse = (size_t)(((d.R + 255) & ~255) + 256) * sizeof(int) + (2 + 64) * SP_WAVES * sizeof(int) +
                         (size_t)SP_WAVES * (std::max(d.Dtop, d.Ein) + 4) * sizeof(float);
    }
    if (ns > 0) DA(m->d_ST, (size_t)gl * ns);
    d.ST = m->d_ST;
    // dense-gradient tile table
    {
        std::vector<DenseTile> tiles;
        const int DTE = 32;
        for (int l = 0; l < L; ++l) {
            const int D = d.D[l], IN = d.IN[l];
            auto add = [&](const float* x0, const float* x1, int ldx, int nrows, int ncols, int coff, int ldo, long long base) {
                for (int r = 0; r < nrows; r += DTE)
                    for (int c = 0; c < ncols; c += DTE) {
                        DenseTile t;
                        t.X0 = x0; t.X1 = x1; t.dV = d.dV[l]; t.base = base; t.ldx = ldx; t.ldv = 3 * D; t.nrows = nrows;
                        t.ncols = ncols; t.coff = coff; t.ldo = ldo; t.r0 = r; t.c0 = c; t.gather = (x0 == nullptr && nrows > 1) ? 1 : 0; t.pad = 0;
                        tiles.push_back(t);
                    }
            };
            const float* yin = (l == 0) ? nullptr : d.hd[l - 1];     // layer 0: gathered in the kernel
            add(yin, yin, IN, IN, 3 * D, 0, 3 * D, d.offWx[l]);                   // dWx  = yin^T dV
            add(d.Hr[l], d.Hr[l], D, D, D, 0, D, d.offWh[l]);                     // dWh  = (H r)^T dV[:, :D]
            add(d.H[l][0], d.H[l][1], D, D, 2 * D, D, 2 * D, d.offWrz[l]);        // dWrz = H^T dV[:, D:]
            add(nullptr, nullptr, 0, 1, 3 * D, 0, 3 * D, d.offBh[l]);             // dBh  = colsum(dV)
        }
        m->ntiles = (int)tiles.size();
        DA(m->d_tiles, tiles.size());
        if (hipMemcpyAsync(m->d_tiles, tiles.data(), tiles.size() * sizeof(DenseTile), hipMemcpyHostToDevice, m->stream) != hipSuccess) {
            g4r_destroy(m); return fail("tile upload");
        }
        if (hipStreamSynchronize(m->stream) != hipSuccess) { g4r_destroy(m); return fail("sync"); }
    }
    // wide layers: the K-sliced kernels of g4r_wide_kernels.cuh.  G4R_WIDE2 (read per model: tests and A/B runs toggle it between
    // models) is a bit mask -- 1 k_gru_p1s + k_gru_gate, 8 k_gru_bwd_bw, 16 k_dense_grad2; 0 = the round-1 kernels -- default: the policy
    // below, from the A/B runs of round 5 (profiles/r05_experiments.md):
    //   16  the 64 x 64 dense-gradient tiles as a launch of their own where the dense gradients outweigh the sparse rows
    //       (6 D >= 2 B + n_sample: BASELINE configs[2] yes -- k_update 24.4 us as one launch, 17.7 + 7.5 as two; configs[3] shape no --
    //       20.6 merged, 20.3 + 13.4 apart: there the merged launch overlaps its two roles)
    //    8  dy as K-slice partial sums wherever a consumer adds them up: the lower layer's k_gru_bwd_pre (any layer above an unfused
    //       one); for layer 0 the row-finishing workgroups of k_dense_grad2 (17.5 -> 7.0 us at configs[2]) or, with the merged k_update,
    //       k_finish_rows as a small launch in front of it (configs[3] shape: 10.8 -> 5.0 + 4.2 us, step 170.3 -> 167.7)
    //    1  phase 1 as partial sums + k_gru_gate from D = 512 on (25.0 -> 18.3 + 4.5 us at configs[2]; D = 256: 13.9 -> 12.9 + 4.3, off)
    // K-slice lengths for A/B runs: G4R_P1_KS (<= 128), G4R_BB_KS.
    {
        const int mask_env = env_int("G4R_WIDE2", -1);
        int dmax_ = 0;
        for (int l = 0; l < L; ++l) dmax_ = std::max(dmax_, d.D[l]);
        const bool automask = mask_env < 0;
        const int mask = automask ? (1 | 8 | (6 * dmax_ >= d.R ? 16 : 0)) : mask_env;
        const int nrt = cdiv(B, 64);
        const bool wdense = (mask & 16) && wide_layer(dmax_) && !(cfg->embed_mode == G4R_EMBED_ONEHOT);
        size_t dyp_floats = 0, vp_floats = 0;
        for (int l = 0; l < L; ++l) {
            const int D = d.D[l], IN = d.IN[l];
            g4r_model::WideGeo& G = m->wg[l];
            const bool ok = wide_layer(D) && D % 64 == 0 && IN % 16 == 0 && IN >= 64 && !(l == 0 && cfg->embed_mode == G4R_EMBED_ONEHOT);
            if (!ok) continue;
            // phase 1: slices of <= 128 units (the whole slice of a workgroup is in flight at once: gemm_tile2k_full); k_gru_gate adds
            // up <= 8 input slices / <= 16 slices in all
            if ((mask & 1) && (!automask || D >= 512)) {
                int ks = std::min(128, std::max(16, env_int("G4R_P1_KS", 128) / 16 * 16));
                G.ny = cdiv(IN, ks); G.kys = ((cdiv(IN, G.ny) + 15) / 16) * 16; G.ny = cdiv(IN, G.kys);
                G.nh = cdiv(D, ks); G.khs = ((cdiv(D, G.nh) + 15) / 16) * 16; G.nh = cdiv(D, G.khs);
                if (G.ny <= 8 && G.ny + G.nh <= 16) {
                    G.use |= 1;
                    vp_floats = std::max(vp_floats, (size_t)(G.ny + G.nh) * B * 3 * D);
                }
            }
            // dy: enough slices of >= 128 (multiples of 32) to give every CU a workgroup, <= 16 (what the consumers add up in one round trip)
            const bool consumer = (l == 0) ? true : (!fused_bwd(d, l - 1) || lean_gru(d, l - 1));      // (layer 0: the row-finishing workgroups of k_dense_grad2, or k_finish_rows in front of the merged k_update)
            if ((mask & 8) && consumer) {
                const int K = 3 * D, tiles = cdiv(IN, 64) * nrt, forced = env_int("G4R_BB_KS", 0);
                int n = std::min(std::max(1, cdiv(m->n_cu, std::max(tiles, 1))), std::max(1, K / 128));
                int ks = ((cdiv(K, n) + 31) / 32) * 32;
                if (forced > 0) ks = std::max(32, forced / 32 * 32);
                if (cdiv(K, ks) <= 16) {
                    G.use |= 8; G.bbk = ks; G.bbn = cdiv(K, ks);
                    d.bbn[l] = G.bbn;
                    dyp_floats = std::max(dyp_floats, (size_t)G.bbn * B * IN);
                }
            }
        }
        if (dyp_floats) DA(d.dyp, dyp_floats);
        if (vp_floats) DA(d.vp, vp_floats);
        m->wide_dense = wdense;
        if (m->wide_dense) {
            std::vector<DenseTile> tiles;
            for (int l = 0; l < L; ++l) {
                const int D = d.D[l], IN = d.IN[l];
                auto add = [&](const float* x0, const float* x1, int ldx, int nrows, int ncols, int coff, int ldo, long long base) {
                    for (int r = 0; r < nrows; r += 64)
                        for (int c = 0; c < ncols; c += 64) {
                            DenseTile t;
                            t.X0 = x0; t.X1 = x1; t.dV = d.dV[l]; t.base = base; t.ldx = ldx; t.ldv = 3 * D; t.nrows = nrows;
                            t.ncols = ncols; t.coff = coff; t.ldo = ldo; t.r0 = r; t.c0 = c; t.gather = (x0 == nullptr && nrows > 1) ? 1 : 0; t.pad = 0;
                            tiles.push_back(t);
                        }
                };
                const float* yin = (l == 0) ? nullptr : d.hd[l - 1];
                add(yin, yin, IN, IN, 3 * D, 0, 3 * D, d.offWx[l]);
                add(d.Hr[l], d.Hr[l], D, D, D, 0, D, d.offWh[l]);
                add(d.H[l][0], d.H[l][1], D, D, 2 * D, D, 2 * D, d.offWrz[l]);
                add(nullptr, nullptr, 0, 1, 3 * D, 0, 3 * D, d.offBh[l]);      // nrows == 1: the column-sum role
            }
            m->ntiles64 = (int)tiles.size();
            DA(m->d_tiles64, tiles.size());
            if (hipMemcpyAsync(m->d_tiles64, tiles.data(), tiles.size() * sizeof(DenseTile), hipMemcpyHostToDevice, m->stream) != hipSuccess ||
                hipStreamSynchronize(m->stream) != hipSuccess) { g4r_destroy(m); return fail("tile upload"); }
        }
    }
#undef DA
    // LDS opt-in
    m->smem_score = ((size_t)(SC_BM + 32) * (SC_KC + 2) + 32) * sizeof(float);
    m->smem_loss = (size_t)(2 * d.ldSc + 18 * LOSS_NW) * sizeof(float);
    m->loss_long = m->smem_loss > (size_t)(156 * 1024);      // one row copy in LDS, the other in the score row itself (k_loss_rows<true>)
    if (m->loss_long) m->smem_loss = (size_t)(d.ldSc + 18 * LOSS_NW) * sizeof(float);
    // four columns per thread and trip from 4 columns per thread on (measured, round 5: B = 512 with 8192 negatives 15.2 -> 13.0 us;
    // 2176 / 2528 columns: 4.7 / 4.4 us either way)
    m->loss_quads = d.ldSc >= 4 * LOSS_T;
    const int big = 156 * 1024;      // leaves room for the few bytes of static LDS some kernels use (__syncthreads_or)
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p1_n32, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p1_n64, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p2_w4, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p2_w8d, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_k128, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_k64, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_t2, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_t3, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_mt_4s, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_bmt, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_bwd_n, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_bwd_fused, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_fwd_fused, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_bwd_w, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_bwd2, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_bwd_a_w4, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_bwd_a_w8d, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_bwd_b, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_dense_grad<32>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update_generic<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update_generic<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_sparse_update_generic<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_update<1, 32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_update<1, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_update<2, 32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_update<2, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_store, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_count, hipFuncAttributeMaxDynamicSharedMemorySize, big));
#define G4R_LOSS_ATTR(L, S)                                                                                                    \
    if (!L) HIPCHK(hipFuncSetAttribute((const void*)k_loss_rows<false, S, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big)); \
    HIPCHK(hipFuncSetAttribute((const void*)k_loss_rows<L, S, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, big))
    G4R_LOSS_ATTR(false, 0); G4R_LOSS_ATTR(false, 1); G4R_LOSS_ATTR(false, 2); G4R_LOSS_ATTR(false, 3);
    G4R_LOSS_ATTR(true, 0); G4R_LOSS_ATTR(true, 1); G4R_LOSS_ATTR(true, 2); G4R_LOSS_ATTR(true, 3);
#undef G4R_LOSS_ATTR
    if (m->smem_loss > (size_t)big) { g4r_destroy(m); return fail("batch_size + n_sample too large for the row-loss kernel (one copy of a score row must fit the 160 KB of LDS)"); }
    if (m->smem_sparse > (size_t)big) { g4r_destroy(m); return fail("2 * batch_size + n_sample too large for the sparse update (the step's list of gathered rows must fit the 160 KB of LDS)"); }
    if (m->exact) {
        const size_t rlist = d.xmode == 3 ? (size_t)d.xn * 2 * B + d.ns : (size_t)d.R * d.xn;      // xlist_len: entries of the exchanged list
        m->smem_exact = (size_t)(((rlist + 255) & ~(size_t)255) + 256) * sizeof(int) + 64 * SP_WAVES * sizeof(int);
        if (m->smem_exact > (size_t)big) {
            g4r_destroy(m);
            return fail("sparse_exact: the exchanged occurrence list (REDUCE form: nranks * 2 * batch_size + n_sample entries; MEAN / SUM: nranks * (2 * batch_size + n_sample)) does not fit the 160 KB of LDS the update stages it in -- use the GPU-local mode (sync_every) at this shape");
        }
    }
    { float* z = nullptr; if (dalloc(m, &z, ZROW_FLOATS)) { g4r_destroy(m); return -1; } d.zrow = z; }
    if (getenv("G4R_CLK")) {
        if (dalloc(m, &d.dbgclk, 64 + 8 * (size_t)d.R) || dalloc(m, &d.dbgtile, 8 * (size_t)(4096 + 4096))) { g4r_destroy(m); return -1; }
    }
    if (build_lean_args(m)) { g4r_destroy(m); return -1; }
    if (dalloc(m, &m->d_dm, 1) || sync_dm(m)) { g4r_destroy(m); return -1; }
    if (m->d_leanU) {
        m->h_leanU.mp = m->d_dm;
        if (hipMemcpyAsync(m->d_leanU, &m->h_leanU, sizeof(LeanU), hipMemcpyHostToDevice, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) { g4r_destroy(m); return fail("lean args upload"); }
    }
    if (m->d_leanS) {
        m->h_leanS.mp = m->d_dm;
        if (hipMemcpyAsync(m->d_leanS, &m->h_leanS, sizeof(LeanS), hipMemcpyHostToDevice, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) { g4r_destroy(m); return fail("lean args upload"); }
    }
    *out = m;
    return 0;
}

void g4r_destroy(g4r_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->cfg.device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
    if (m->gexec) (void)hipGraphExecDestroy(m->gexec);
    if (m->gexec_small) (void)hipGraphExecDestroy(m->gexec_small);
    if (m->gexec_head) (void)hipGraphExecDestroy(m->gexec_head);
    if (m->comm_ready) (void)ncclCommDestroy(m->comm);
    for (void* q : m->p2p_peer) if (q) (void)hipIpcCloseMemHandle(q);
    if (m->p2p_region) (void)hipFree(m->p2p_region);
    for (auto e : m->evs) (void)hipEventDestroy(e);
    for (auto e : m->ev_df) if (e) (void)hipEventDestroy(e);
    for (g4r_model::Scratch* sc : {&m->sc_ids, &m->sc_blk, &m->sc_cnt, &m->sc_all, &m->sc_send, &m->sc_pack, &m->sc_recv, &m->sc_hall}) {
        if (sc->p) { if (sc->host) (void)hipHostFree(sc->p); else (void)hipFree(sc->p); }
        sc->p = nullptr; sc->cap = 0;
    }
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->comm_stream) (void)hipStreamDestroy(m->comm_stream);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}
