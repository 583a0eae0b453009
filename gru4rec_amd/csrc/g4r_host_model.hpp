// g4r_host_model.hpp -- part of libgru4rec_hip.so's host code; included once, by g4r_api.hip (one translation unit: the kernels are templates
// instantiated there).  Holds: error string, kernel time slots, the model handle (g4r_model), allocation helpers, launch geometries and the shape policies that pick a kernel.
static thread_local std::string g_err;
static int fail(const std::string& s) { g_err = s; return -1; }
// printf-style setter for the host-only translation units of the library (g4r_io.cpp)
void g4r_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
#define HIPCHK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess)                                                                            \
            return fail(std::string(#x) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)
#define NCCLCHK(x)                                                                                         \
    do {                                                                                                   \
        ncclResult_t e_ = (x);                                                                             \
        if (e_ != ncclSuccess)                                                                             \
            return fail(std::string(#x) + ": " + ncclGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)

enum { KN_GRU_P1 = 0, KN_GRU_P2, KN_SCORE_FWD, KN_LOSS, KN_SCORE_BWD, KN_BWD_PRE, KN_BWD_A, KN_BWD_B, KN_DENSE, KN_ALLREDUCE,
       KN_DENSE_APPLY, KN_SPARSE, KN_UPDATE, KN_BWD_FUSED, KN_FWD_FUSED, KN_GATE, KN_FLUSH, KN_SCAN, KN_FINISH, KN_GRU_V, KN_GRU_H, KN_GRU_DA, KN_GRU_DY, KN_COUNT };
static const char* KN_NAMES[KN_COUNT] = {"k_gru_p1", "k_gru_p2", "k_score_fwd", "k_loss_rows", "k_score_bwd", "k_gru_bwd_pre",
                                         "k_gru_bwd_a", "k_gru_bwd_b", "k_dense_grad", "rccl_allreduce", "k_dense_apply",
                                         "k_sparse_update", "k_update", "k_gru_bwd", "k_gru_fwd", "k_gru_gate", "k_sparse_flush", "k_defer_scan", "k_finish_rows", "k_gru_v", "k_gru_h", "k_gru_da", "k_gru_dy"};

struct EvRec { int kn; hipEvent_t a, b; };

struct g4r_model {
    g4r_config cfg;
    DevModel dm;                 // host master copy of the device-resident model descriptor
    DevModel* d_dm = nullptr;    // what the kernels read (passed by pointer: 8-byte kernarg)
    int n_cu = 256;              // compute units of the device (tile-count heuristics)
    int p2_geo_env = -1, ba_geo_env = -1;      // G4R_P2_GEO / G4R_BA_GEO at g4r_create (-1: deep_geometry's policy)
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;           // all-reduce + dense Adagrad next to the sparse update (nranks > 1)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<void*> allocs;
    // plan
    int *d_in = nullptr, *d_out = nullptr, *d_M = nullptr, *d_cmaps = nullptr;
    unsigned char* d_reset = nullptr;
    float* d_loss = nullptr;
    int64_t T = 0, loss_cap = 0;
    std::vector<int64_t> compact_steps;
    // samples
    int* d_ST = nullptr;
    float *d_P = nullptr, *d_lqt = nullptr, *d_lqs = nullptr;
    int64_t gl = 0;
    bool store_frozen = false, have_pop = false;
    unsigned refills = 0;
    int64_t gstep = 0;
    // launch geometry
    DenseTile* d_tiles = nullptr;
    int ntiles = 0, nblkA = 0, nblkB = 0, ndtA = 0, ndtB = 0, nrtB = 0, nblk_occ = 0, nblk_occ_g = 0;
    size_t smem_score = 0, smem_loss = 0, smem_sparse = 0;
    bool loss_long = false;      // k_loss_rows<true>: score rows too long for two LDS copies
    bool loss_quads = false;     // k_loss_rows<., ., 4>: four columns per thread and trip (long score rows)
    // wide layers (g4r_wide_kernels.cuh): per layer which kernels run (bit 1 k_gru_p1s + k_gru_gate, 8 k_gru_bwd_bw) and their K-slice
    // geometry; wide_dense: the 64 x 64 dense-gradient tiles (k_dense_grad2, mask bit 16) as a launch of their own for the whole model
    struct WideGeo { int use = 0, ny = 1, nh = 1, kys = 0, khs = 0, bbn = 1, bbk = 0; };
    WideGeo wg[G4R_MAX_LAYERS];
    bool wide_dense = false;
    bool lean_upd = true;        // k_update_l allowed (G4R_LEAN_UPDATE=0 at create: the merged k_update, as the deferred mode runs it -- the reference run of tests/test_gpu_defer.py)
    bool defer_on = false;       // deferred row updates (k_defer_scan / k_sparse_flush around every replay of the step graph)
    bool defer_broken = false;   // a call failed between a window's scan and its flush: pending row updates were lost, the handle refuses to go on
    hipEvent_t ev_df[4] = {nullptr, nullptr, nullptr, nullptr};      // profiling: scan / flush launches of a window
    DenseTile* d_tiles64 = nullptr;
    int ntiles64 = 0;
    float* d_tmpH = nullptr;
    // graph
    LeanS h_leanS;
    std::vector<LeanV> h_leanV; std::vector<LeanH> h_leanH; std::vector<LeanDa> h_leanDa; std::vector<LeanDy> h_leanDy;
    LeanS* d_leanS = nullptr;      // argument block of k_score_s
    LeanB* d_leanB = nullptr; LeanB h_leanB;      // argument block of k_score_b
    LeanU* d_leanU = nullptr; LeanU h_leanU;      // argument block of k_update_l, its 16 x 64 dense tiles
    DenseTile* d_tiles16 = nullptr; int ntiles16 = 0;
    LeanV* d_leanV = nullptr; LeanH* d_leanH = nullptr; LeanDa* d_leanDa = nullptr; LeanDy* d_leanDy = nullptr;      // [layers] argument blocks (g4r_lean_kernels.cuh)
    hipGraphExec_t gexec = nullptr;
    hipGraphExec_t gexec_small = nullptr;        // single GPU: G4R_GRAPH_STEPS_SMALL steps, for what a run leaves after the big replays
    hipGraphExec_t gexec_head = nullptr;         // N > 1 fallback: one step's kernels up to the dense gradients, RCCL eager behind it
    int graph_steps = 0;
    bool dist_graph_failed = false;              // capturing the step with its RCCL all-reduce did not work: head graph + eager tail
    // profiling
    bool profiling = false;
    bool profile_split = false;
    bool exact = false;                          // g4r_config::sparse_exact with nranks > 1
    size_t smem_exact = 0;
    double kn_ms[KN_COUNT] = {0};
    int64_t kn_n[KN_COUNT] = {0};
    std::vector<hipEvent_t> evs;
    // prediction
    int pbatch = 0, ppar = 0;
    float* pH[G4R_MAX_LAYERS][2] = {{nullptr}};
    float* phout[G4R_MAX_LAYERS] = {nullptr};
    float *pVc[G4R_MAX_LAYERS] = {nullptr}, *pz[G4R_MAX_LAYERS] = {nullptr}, *pHr[G4R_MAX_LAYERS] = {nullptr};
    int *p_in = nullptr, *p_items = nullptr, *p_tgt = nullptr, *p_keep = nullptr;
    unsigned char* p_zero = nullptr;
    float *p_scores = nullptr, *p_ranks = nullptr;
    int* p_cnt = nullptr;                        // [pbatch][2] streamed (greater, equal) counts of the evaluation
    int64_t p_scores_cap = 0, p_items_cap = 0, p_nsel = 0, p_ldo = 0;
    unsigned tie_ctr = 0;                        // evaluation step counter of the 'tiebreaking' noise stream
    // rccl
    ncclComm_t comm = nullptr;
    bool comm_ready = false;
    // one-shot all-reduce of the dense gradients through peer memory (g4r_p2p_*): this rank's exchange region, the peers' regions
    // as mapped here (IPC), the kernel's argument block
    bool p2p_ready = false;
    void* p2p_region = nullptr;
    void* p2p_peer[G4R_P2P_MAX] = {nullptr};
    unsigned* p2p_round = nullptr;
    int p2p_nblk = 0, p2p_cap = 0;
    P2PArgs p2p_args;
    bool virtual_ranks = false;                  // member of a g4r_virtual_train_steps group: the dense gradients are summed in process
    float* d_vsum = nullptr;                     // scratch of that sum (first member of the group)
    // reconciliation of the GPU-local item tables (g4r_sync_kernels.cuh): per table group (0: Wy / By rows, 1: E rows) the
    // planes (current values, common base, row width) and scratch
    struct SyncPlane { float* cur; float* base; int W; int kind; };      // kind: 0 parameter / velocity, 1 optimizer statistic
    std::vector<SyncPlane> planes[2];
    int sync_rule[2] = {G4R_SYNC_MEAN, G4R_SYNC_SUM};      // combine rule of the parameter planes / of the statistic planes
    bool sync_rule_user = false;                           // set through g4r_sync_set_rule: g4r_sync_enable keeps it
    unsigned char* d_touched = nullptr;
    unsigned char* d_rowcnt = nullptr;           // [n_items] scratch: number of parts that hold a row (MEAN rule)
    int sync_every_dev = 0;                      // > 0: g4r_train_steps reconciles the (dense-form) item tables itself every that many steps
    int64_t since_sync = 0, n_dev_syncs = 0;
    // scratch of the packed-parts reconciliation, kept between calls (a call used to pay five hipMalloc / hipFree pairs)
    struct Scratch { void* p = nullptr; size_t cap = 0; bool host = false; };
    Scratch sc_ids, sc_blk, sc_cnt, sc_all, sc_send, sc_pack, sc_recv, sc_hall;      // sc_hall: pinned host copy of the gathered id lists
    float* d_dense[2] = {nullptr, nullptr};      // dense reconciliation buffers [n_items][sum of plane widths + 1] per table group (small catalogues)
    bool sync_on = false;
};

template <class T>
static int dalloc(g4r_model* m, T** p, size_t n, bool zero = true) {
    void* q = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc(&q, n * sizeof(T)));
    if (zero) HIPCHK(hipMemsetAsync(q, 0, n * sizeof(T), m->stream));
    m->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}
static void dfree(g4r_model* m, void* p) {
    if (!p) return;
    auto it = std::find(m->allocs.begin(), m->allocs.end(), p);
    if (it != m->allocs.end()) m->allocs.erase(it);
    (void)hipFree(p);
}
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static constexpr auto k_score_store = k_score_all<32, false>;     // scores -> memory
static constexpr auto k_score_count = k_score_all<32, true>;      // scores compared with the row's target on the fly

static inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
// dynamic LDS of the tile-GEMM kernels (g4r_gemm.cuh)
template <int BM, int BN, int BK, bool AKM, bool BNK>
static constexpr size_t tile_smem() { return (size_t)TileCfg<BM, BN, BK, AKM, BNK>::SMEM_FLOATS * sizeof(float); }
static const size_t SMEM_NN = tile_smem<GT_BM, GT_BN, GT_BK, false, false>() + GT_BM * sizeof(int);   // A [m][k], B [k][n] (+ row items)
static const size_t SMEM_NT = tile_smem<GT_BM, GT_BN, GT_BK, false, true>() + GT_BM * sizeof(int);    // A [m][k], B [n][k] (+ row items)
static const size_t SMEM_TN = tile_smem<GT_BM, GT_BN, GT_BK, true, false>();    // A [k][m], B [k][n]
// wide layers: 64-column tiles halve the number of GRU phase-1 workgroups (all resident at once) and read the weights in
// 256-byte runs; the 32-column tiles spread the tiny GEMMs of D ~ 100 over more CUs
static constexpr auto k_gru_p1_n32 = k_gru_p1<GT_BN, P1_BK>;
static constexpr auto k_gru_p1_n64 = k_gru_p1<64, 256>;
static const size_t SMEM_P1_N64 = tile_smem<GT_BM, 64, 256, false, false>() + GT_BM * sizeof(int);
// GRU forward / backward of a narrow layer as the four lean launches of g4r_lean_kernels.cuh (k_gru_v, k_gru_h / k_gru_da, k_gru_dy): training,
// in and D up to LN_MAXD.  G4R_NO_LEAN=1 (read once): the fused single-launch kernels of round 2 instead (A/B runs, tests of both forms).
static inline bool lean_gru(const DevModel& d, int l) {
    static const bool off = getenv("G4R_NO_LEAN") != nullptr;
    return !off && d.D[l] <= LN_MAXD && d.IN[l] <= LN_MAXD && d.D[l] % 4 == 0 && d.IN[l] % 4 == 0 && d.IN[l] >= 4 &&
           !(l == 0 && d.embed_mode == G4R_EMBED_ONEHOT);
}
// GRU backward in one launch (k_gru_bwd_fused) for layers whose operands fit its LDS plan
static inline bool fused_bwd(const DevModel& d, int l) {
    return d.D[l] <= BF_MAXD && d.D[l] % 4 == 0 && d.IN[l] % 4 == 0 && !(l == 0 && d.embed_mode == G4R_EMBED_ONEHOT);
}
// GRU forward in one launch (k_gru_fwd_fused) for layers whose weights fit its LDS plan (in + D up to ~200)
static inline bool fused_fwd(const DevModel& d, int l) {
    return d.D[l] <= FF_LDR && d.IN[l] <= FF_LDR && d.D[l] % 4 == 0 && d.IN[l] % 4 == 0 && d.IN[l] >= 4 &&      // its load maps cover 112 rows / columns
           !(l == 0 && d.embed_mode == G4R_EMBED_ONEHOT) && (size_t)fwd_fused_lds(d.IN[l], d.D[l]).total * sizeof(float) <= 156 * 1024;
}
static inline size_t smem_fused_bwd(int D) { return (size_t)((((BF_ROWS + 32) * (3 * D + 2) + D * (D + 2) + 32 + 3) & ~3) + 4 * 6 * 64) * sizeof(float); }
static inline bool wide_layer(int D) { return D >= 256; }
static const size_t SMEM_P1 = tile_smem<GT_BM, GT_BN, P1_BK, false, false>() + GT_BM * sizeof(int);
// k_gru_p2 / k_gru_bwd_a (32 x 32 tiles over K = D): 4 waves and 128-deep chunks; where the launch leaves CUs idle and K is longer than
// two such chunks, 8 waves (two wave groups that split every chunk's k range) and 256-deep chunks -- one workgroup per CU either way, half
// the memory round trips and half the MFMA chain per tile.  Measured (round 5, us): B = 240, D = 512: k_gru_p2 9.7 -> 8.2, k_gru_bwd_a
// 7.3 -> 6.2; B = 512, D = 256: 6.7 -> 6.4 / 4.5 -> 4.35 (left on the 4-wave form); 8 waves x 128 (9.2) and, for k_gru_bwd_a, 8 waves x
// 512 = the whole K in one chunk (6.4) were no better.  G4R_P2_GEO / G4R_BA_GEO = 0 / 1 override (tests).
static constexpr auto k_gru_p2_w4 = k_gru_p2<GT_NTH, GT_BK>;
static constexpr auto k_gru_p2_w8d = k_gru_p2<512, 256>;
static const size_t SMEM_P2_256 = tile_smem<GT_BM, GT_BN, 256, false, false>() + GT_BM * sizeof(int);
static constexpr auto k_gru_bwd_a_w4 = k_gru_bwd_a<GT_NTH, GT_BK>;
static constexpr auto k_gru_bwd_a_w8d = k_gru_bwd_a<512, 256>;
static const size_t SMEM_BA_256 = tile_smem<GT_BM, GT_BN, 256, false, true>();
static inline int deep_geometry(int forced, int n_cu, int D, int rows) {
    if (forced >= 0) return forced != 0;
    return D >= 384 && cdiv(D, GT_BN) * cdiv(rows, GT_BM) <= n_cu;
}
static const size_t SMEM_BB = tile_smem<GT_BM, GT_BN, BB_BK, false, true>() + GT_BM * sizeof(int);
static constexpr auto k_score_fwd_k128 = k_score_fwd<GT_BN, GT_BK>;
// long score rows: 64-deep K chunks (more resident workgroups).  Measured at B = 512, N = 8704, D = 256 (us): 64 x 32 tiles
// with K chunks of 64: 39.2, 64 x 64 / 64: 41.4, 64 x 64 / 128: 42.4, 64 x 64 / 32: 45.9 -- the tile shape is not what bounds it
#define SFW_BN 32
#define SFW_BK 64
static constexpr auto k_score_fwd_k64 = k_score_fwd<SFW_BN, SFW_BK>;
static constexpr auto k_score_fwd_t2 = k_score_fwd<64, 32, T2_BK>;      // gemm_tile2: 64 x 64 tiles, double-buffered T2_BK-deep chunks
static const size_t SMEM_SF2 = (size_t)Tile2Cfg<T2_BK>::SMEM_FLOATS * sizeof(float);
static constexpr auto k_score_fwd_t3 = k_score_fwd<64, 32, 3>;          // gemm_tile3: the same tile fed by LDS-DMA through a ring of stages
static const size_t SMEM_SF3 = (size_t)Tile3Cfg<T3_NST, T3_BKS>::SMEM_FLOATS * sizeof(float);
static inline bool score_tile2() { return true; }
static inline bool wide_scores(const DevModel& d);
// gemm_tile2k scoring backward (k_score_bwd2): long score rows / big batches and D a multiple of 64
static inline bool score_bwd2(const DevModel& d) { return wide_scores(d) && score_tile2() && d.Dtop % 64 == 0; }
// macro-tile scoring backward (k_score_bmt, g4r_score_bmt.cuh): role A in 272 x 32 tiles, role B in 64 x 128 tiles x `ks` slabs, as many
// of each and together two per CU.  Returns ks (0: k_score_bwd2).  G4R_NO_BMT=1: off (A/B runs).
static inline int score_bmt_slabs(const DevModel& d, int n_cu) {
    static const bool off = getenv("G4R_NO_BMT") != nullptr;
    if (off || !score_bwd2(d) || d.Dtop % 128 != 0 || d.Dtop > 512 || d.ldSc % BMT_WA != 0 || d.ldSc > 0xFFFF || d.B > 0xFFFF) return 0;
    if (lean_gru(d, d.n_layers - 1) || fused_bwd(d, d.n_layers - 1)) return 0;      // (the Adagrad rule of the item rows rides on the top layer's k_gru_bwd_a)
    const int ntA = d.ldSc / BMT_WA * (d.Dtop / 32), den = cdiv(d.B, 64) * (d.Dtop / 128);
    if (ntA % den != 0 || ntA % 8 != 0 || ntA > n_cu || ntA * 8 < n_cu * 7 || BMT_WA % (d.Dtop / 32) != 0 || 4 * (BMT_WA / (d.Dtop / 32)) > 256) return 0;      // (bias columns: 272 / (D / 32) per tile, four threads each)
    const int ks = ntA / den;
    if (ks < 2 || ks > 24 || d.ldSc % ks != 0 || (d.ldSc / ks) % 32 != 0 || d.ldSc / ks > 2048) return 0;
    return ks;
}
static const size_t SMEM_BMT = std::max((size_t)BMT_NST_A * BMT_STAGE_A * sizeof(float), (size_t)BMT_NST_B * BMT_STAGE_B * sizeof(float) + 2048 * sizeof(int));
static const size_t SMEM_SF64 = tile_smem<SF_BM, SFW_BN, SFW_BK, false, true>() + SFW_BN * sizeof(int);
static constexpr auto k_score_bwd_n = k_score_bwd<32, GT_BK>;
static constexpr auto k_score_bwd_w = k_score_bwd<64, 64>;
static const size_t SMEM_SBW = std::max(tile_smem<64, 64, 64, true, false>(), tile_smem<64, 64, 64, false, false>());
static inline bool wide_scores(const DevModel& d) {
    const bool off = false;
    const int minB = 256, minN = 4096;
    // (a top layer that is a multiple of 64 takes the 64 x 64 tiles of k_score_bwd2 from B = 192, 2048 columns on: B = 240, N = 2288,
    // D = 512 measured 22.2 vs 25.1 us against the 32 x 32 tiles)
    const int d64 = 1;
    return !off && ((d.B >= minB && d.ldSc >= minN) || (d64 && d.Dtop % 64 == 0 && d.B >= std::min(minB, 192) && d.ldSc >= std::min(minN, 2048)));
}
// LDS-DMA tiles (gemm_tile3, k_score_fwd_t3), D a multiple of 32: where gemm_tile2 served (long score rows / big batches), and
// for a wide top layer (D >= 256) whenever the batch fills 64-row tiles -- there the launch is a few hundred tiles, fewer than
// the chip holds at once, and only the ring's depth hides a stage's memory round trip (B = 240, N = 2288, D = 512: 18.7 -> 15.0 us)
#define ZROW_FLOATS 8192      // DevModel::zrow: an LDS-DMA tile walks K floats along it
#define G4R_DEFER_SLOTS 16    // ring slots of the step planes = steps of a deferral window (= G4R_GRAPH_STEPS; a power of two)
static inline bool score_fwd_dma(const DevModel& d) {
    if (d.Dtop % 32 != 0) return false;
    return wide_scores(d) || (d.Dtop >= 256 && d.B >= 64 && d.ldSc >= 1024);
}
static const size_t SMEM_SF = tile_smem<SF_BM, GT_BN, GT_BK, false, true>() + GT_BN * sizeof(int);
// macro-tile scoring forward (k_score_mt, g4r_score_mt.cuh): the score matrix cut into at most n_cu tiles of 64 rows x W = 64 NB + 16
// columns, one per compute unit.  Applies when that W is one of the instantiated widths: returns W (0: the 64 x 64 tiles).
// G4R_NO_MT=1: off (A/B runs).
static constexpr auto k_score_mt_4s = k_score_mt<4, true>;       // W = 272: B = 512, N = 8704 on 256 CUs
static const size_t SMEM_MT_4S = (size_t)MtCfg<4, true>::SMEM_FLOATS * sizeof(float);
static inline int score_mt_width(const DevModel& d, int n_cu) {
    static const bool off = getenv("G4R_NO_MT") != nullptr;
    if (off || !score_fwd_dma(d) || d.Dtop % 16 != 0) return 0;
    const int nrb = cdiv(d.B, 64), G = n_cu / nrb;
    if (G < 1) return 0;
    const int W = 16 * cdiv(d.ldSc, 16 * G);
    return (W == 272 && nrb * cdiv(d.ldSc, W) * 8 >= n_cu * 7) ? W : 0;      // (tiles for >= 7/8 of the CUs)
}
// scoring forward as register-fed 32 x 32 tiles (k_score_s, g4r_lean_kernels.cuh): narrow top layers at RSC15-like sizes, i.e. where the
// LDS-staged 64 x 32 tiles of k_score_fwd ran (neither the wide-score nor the LDS-DMA tiles apply).  G4R_NO_LEAN=1: off.
static inline bool lean_scores(const DevModel& d) {
    static const bool off = getenv("G4R_NO_LEAN") != nullptr;
    return !off && d.Dtop <= LN_MAXD && d.Dtop % 4 == 0 && !wide_scores(d) && !score_fwd_dma(d) && d.B < 65536 && d.ldSc < 65536;
}
// ... and the scoring backward as k_score_b (eight waves over a K of <= 128 batch rows / 128-column slabs)
static inline bool lean_score_bwd(const DevModel& d) { return lean_scores(d) && d.B <= 128; }
static const size_t SMEM_T2K = (size_t)(4 * 64 * 16) * sizeof(float);                               // gemm_tile2k: two 16-deep buffers per operand
static const size_t SMEM_T3 = (size_t)Tile3Cfg<3, 32>::SMEM_FLOATS * sizeof(float);                 // gemm_tile3: ring of three 32-deep stages
// publish the host descriptor to the device copy (stream-ordered; pageable source is staged before return)
static int sync_dm(g4r_model* m) {
    HIPCHK(hipMemcpyAsync(m->d_dm, &m->dm, sizeof(DevModel), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}
