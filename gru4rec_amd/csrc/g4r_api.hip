// libgru4rec_hip.so -- host side of the C ABI declared in include/gru4rec_hip.h.
// Owns device memory, the HIP stream, the captured step graph and the (optional) RCCL communicator.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <string>
#include <vector>

#include "g4r_eval_kernels.cuh"
#include "g4r_sync_kernels.cuh"
#include "g4r_micro_kernels.cuh"
#include "g4r_wide_kernels.cuh"

// Host code below is compiled in the host pass only: on the device pass the descriptor pointer fields are
// address-space qualified (g4r_device.cuh) and the template kernels are instantiated explicitly.
#if !defined(__HIP_DEVICE_COMPILE__)

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
       KN_DENSE_APPLY, KN_SPARSE, KN_UPDATE, KN_BWD_FUSED, KN_FWD_FUSED, KN_GATE, KN_FLUSH, KN_SCAN, KN_FINISH, KN_COUNT };
static const char* KN_NAMES[KN_COUNT] = {"k_gru_p1", "k_gru_p2", "k_score_fwd", "k_loss_rows", "k_score_bwd", "k_gru_bwd_pre",
                                         "k_gru_bwd_a", "k_gru_bwd_b", "k_dense_grad", "rccl_allreduce", "k_dense_apply",
                                         "k_sparse_update", "k_update", "k_gru_bwd", "k_gru_fwd", "k_gru_gate", "k_sparse_flush", "k_defer_scan", "k_finish_rows"};

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
    // wide layers (g4r_wide_kernels.cuh): per layer which kernels run (bit 1 k_gru_p1s + k_gru_gate, 8 k_gru_bwd_bw) and their K-slice
    // geometry; wide_dense: the 64 x 64 dense-gradient tiles (k_dense_grad2, mask bit 16) as a launch of their own for the whole model
    struct WideGeo { int use = 0, ny = 1, nh = 1, kys = 0, khs = 0, bbn = 1, bbk = 0; };
    WideGeo wg[G4R_MAX_LAYERS];
    bool wide_dense = false;
    bool defer_on = false;       // deferred row updates (k_defer_scan / k_sparse_flush around every replay of the step graph)
    hipEvent_t ev_df[4] = {nullptr, nullptr, nullptr, nullptr};      // profiling: scan / flush launches of a window
    DenseTile* d_tiles64 = nullptr;
    int ntiles64 = 0;
    float* d_tmpH = nullptr;
    // graph
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
// GRU backward in one launch (k_gru_bwd_fused) for layers whose operands fit its LDS plan
static inline bool fused_bwd(const DevModel& d, int l) {
    static const bool off = getenv("G4R_NO_FUSED_BWD") != nullptr;
    return !off && d.D[l] <= BF_MAXD && d.D[l] % 4 == 0 && d.IN[l] % 4 == 0 && !(l == 0 && d.embed_mode == G4R_EMBED_ONEHOT);
}
// GRU forward in one launch (k_gru_fwd_fused) for layers whose weights fit its LDS plan (in + D up to ~200)
static inline bool fused_fwd(const DevModel& d, int l) {
    static const bool off = getenv("G4R_NO_FUSED_FWD") != nullptr;
    return !off && d.D[l] <= FF_LDR && d.IN[l] <= FF_LDR && d.D[l] % 4 == 0 && d.IN[l] % 4 == 0 && d.IN[l] >= 4 &&      // its load maps cover 112 rows / columns
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
static const size_t SMEM_T2K = (size_t)(4 * 64 * 16) * sizeof(float);                               // gemm_tile2k: two 16-deep buffers per operand
static const size_t SMEM_T3 = (size_t)Tile3Cfg<3, 32>::SMEM_FLOATS * sizeof(float);                 // gemm_tile3: ring of three 32-deep stages
// publish the host descriptor to the device copy (stream-ordered; pageable source is staged before return)
static int sync_dm(g4r_model* m) {
    HIPCHK(hipMemcpyAsync(m->d_dm, &m->dm, sizeof(DevModel), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

extern "C" {

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
    DA(d.col_item, d.ldSc); DA(d.cur_in, B); DA(d.cur_col, d.ldSc);
    DA(d.occ_fl, (size_t)(cfg->embed_mode != G4R_EMBED_CONSTRAINED ? 2 : 1) * I * 4);
    DA(d.st, 1);
    // scoring backward geometry: role A tiles (n x d, one spare d column for dSBy), role B tiles (b x d x k-chunk)
    {
        // k_gru_bwd_fused sums the slabs next to everything else it loads: half as many, twice as deep (k_score_bwd +0.4 us at cfg2)
        const int slabs_target = getenv("G4R_KSLABS") ? atoi(getenv("G4R_KSLABS")) : (fused_bwd(d, d.n_layers - 1) ? 9 : 17);
        d.kch = GT_BK * std::max(1, (cdiv(d.ldSc, GT_BK) + slabs_target / 2) / slabs_target);      // ~17 slabs whatever the number of negatives
        if (score_bwd2(d) && !getenv("G4R_KSLABS")) {
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
            const bool consumer = (l == 0) ? true : !fused_bwd(d, l - 1);      // (layer 0: the row-finishing workgroups of k_dense_grad2, or k_finish_rows in front of the merged k_update)
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
    const int big = 156 * 1024;      // leaves room for the few bytes of static LDS some kernels use (__syncthreads_or)
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p1_n32, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p1_n64, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p2_w4, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_gru_p2_w8d, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_k128, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_k64, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_t2, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void*)k_score_fwd_t3, hipFuncAttributeMaxDynamicSharedMemorySize, big));
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
#define G4R_LOSS_ATTR(L, S) HIPCHK(hipFuncSetAttribute((const void*)k_loss_rows<L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, big))
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
    if (dalloc(m, &m->d_dm, 1) || sync_dm(m)) { g4r_destroy(m); return -1; }
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
        dalloc(m, &m->d_reset, (size_t)T * B, false) || dalloc(m, &m->d_M, (size_t)T + 1, true))
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

// ------------------------------------------------------------------------------------------------ the step
static inline bool no_merge_tail() { static const bool v = getenv("G4R_NO_MERGE") != nullptr; return v; }
// float4 chunks per lane a gathered row needs in the sparse update: rows of <= 256 / 512 / 1024 floats
static inline int row_chunks(const DevModel& d) { const int w = std::max(d.Dtop, d.Ein); return w <= 256 ? 1 : (w <= 512 ? 2 : 4); }
// (rows wider than 512 floats take the two-launch form: k_update's register budget is sized for two chunks per lane)
// (wide layers: the dense gradients run as 64 x 64 tiles in a launch of their own, k_dense_grad2, ahead of the sparse row update)
static inline bool merged_update(const g4r_model* m) { return !m->dm.generic && !no_merge_tail() && row_chunks(m->dm) <= 2 && !m->wide_dense; }
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
        if (recs) hipExtLaunchKernelGGL(kern, grid, block, smem, strm, cur_a, cur_b, 0, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kern, grid, block, smem, strm, __VA_ARGS__);                              \
    } while (0)
    const DevModel* dmp = (const DevModel*)m->d_dm;
    StepState* stp = (StepState*)d.st;
    bool merged = false;      // the sparse update already ran inside k_update
    if (part != 2) {
    for (int l = 0; l < L; ++l) {
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
    if (score_fwd_dma(d)) LK(k_score_fwd_t3, dim3(cdiv(d.ldSc, 64), cdiv(B, 64)), dim3(GT_NTH), SMEM_SF3, s, dmp, stp);
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
#define G4R_LK_LOSS(L)                                                                              \
        do {                                                                                        \
            if (spec == 1) LK((k_loss_rows<L, 1>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp);      \
            else if (spec == 2) LK((k_loss_rows<L, 2>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp); \
            else if (spec == 3) LK((k_loss_rows<L, 3>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp); \
            else LK((k_loss_rows<L, 0>), dim3(B), dim3(LOSS_T), m->smem_loss, s, dmp, stp);                \
        } while (0)
        if (m->loss_long) G4R_LK_LOSS(true); else G4R_LK_LOSS(false);
#undef G4R_LK_LOSS
    }
    end();
    begin(KN_SCORE_BWD);
    if (score_bwd2(d)) {
        const int ndt = d.Dtop / 64, nrt = cdiv(B, 64);
        int nA = cdiv(d.ldSc, 64) * ndt, nB = d.ksplit * nrt * ndt, nC = cdiv(d.ldSc, 64);
        LK(k_score_bwd2, dim3(nA + nB + nC), dim3(GT_NTH), (size_t)(4 * 64 * 16) * sizeof(float) + (size_t)std::max(d.kch, 64) * sizeof(int), s, dmp, stp, nA, nB, ndt, nrt);
    } else if (wide_scores(d)) LK(k_score_bwd_w, dim3(m->nblkA + m->nblkB), dim3(GT_NTH), SMEM_SBW + (size_t)d.kch * sizeof(int), s, dmp, stp, m->nblkA, m->ndtA, m->ndtB, m->nrtB);
    else LK(k_score_bwd_n, dim3(m->nblkA + m->nblkB), dim3(GT_NTH), std::max(SMEM_TN, SMEM_NN) + (size_t)d.kch * sizeof(int), s, dmp, stp, m->nblkA, m->ndtA, m->ndtB, m->nrtB);
    end();
    for (int l = L - 1; l >= 0; --l) {
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
            const dim3 ga(cdiv(d.D[l], GT_BN), cdiv(B, GT_BM));
            if (deep_geometry(m->ba_geo_env, m->n_cu, d.D[l], B)) LK(k_gru_bwd_a_w8d, ga, dim3(512), SMEM_BA_256, s, dmp, stp, l);
            else LK(k_gru_bwd_a_w4, ga, dim3(GT_NTH), SMEM_NT, s, dmp, stp, l);
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
    // Optionally the first two run on their own stream next to the sparse update (which touches item rows only)
    // and join before the next step reads the GRU weights
    // (measured on one MI355X with a one-rank communicator: the two cross-stream event dependencies cost ~20 us per
    // step, more than the ~11 us of sparse update they can hide, so the overlap is opt-in: G4R_OVERLAP=1)
    static const bool want_overlap = getenv("G4R_OVERLAP") != nullptr;
    const bool overlap = !d.apply_dense_inplace && !recs && !trace && want_overlap && !d.generic && !merged && !m->p2p_ready && (m->cfg.nranks > 1 || m->comm_ready);
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
// (kernels, RCCL all-reduce, dense apply) with no host work in between; G4R_RCCL_EAGER=1 keeps RCCL out of the graph
// one GPU, staged dense path without a communicator (the generic optimizers: rmsprop / adadelta / adam / plain SGD / grad_cap): no
// collective in the step, so the whole step is captured like the fused single-GPU step (it used to replay a head graph and launch
// its tail eagerly; G4R_NO_LOCAL_GRAPH=1 keeps that)
static inline bool local_staged(const g4r_model* m) {
    return !m->dm.apply_dense_inplace && m->cfg.nranks <= 1 && !m->comm_ready && !m->p2p_ready && !m->virtual_ranks;
}
static inline bool dist_graph_wanted(const g4r_model* m) {
    static const bool eager = getenv("G4R_RCCL_EAGER") != nullptr;
    return !m->dm.apply_dense_inplace && !m->dist_graph_failed &&
           (m->p2p_ready || (m->comm_ready && !eager && !getenv("G4R_OVERLAP")) || local_staged(m));
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
            if (deep_geometry(m->p2_geo_env, m->n_cu, d.D[l], mrows))
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
        if (!getenv("G4R_P2P_COARSE") && hipExtMallocWithFlags(&q, bytes, hipDeviceMallocUncached) == hipSuccess) {
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

}  // extern "C"
#endif  // !__HIP_DEVICE_COMPILE__
