/*
 * gru4rec_hip.h -- C ABI of libgru4rec_hip.so: the MI355X (gfx950) implementation of GRU4Rec's
 * session-parallel mini-batch training / prediction hot path.
 *
 * This is the drop-in boundary.  The reference (hidasib/GRU4Rec) has no FFI of its own: its
 * device code is reached through compiled Theano functions.  Each entry point below replaces one
 * such Theano-function boundary (cited per function as reference file:line).  The Python host
 * (`gru4rec_amd/gru4rec.py`, mirroring the reference's `GRU4Rec` class) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every call returns 0 on success, <0 on error; g4r_last_error() returns the message
 *   - plain pointers + sizes only; the caller owns every host buffer, the library owns device memory
 *   - a model handle is not thread-safe; one HIP stream per handle; calls are synchronous at return
 *     unless stated otherwise
 *   - all matrices are row-major fp32; layer sizes / embedding size must be multiples of 4
 */
#ifndef GRU4REC_HIP_H
#define GRU4REC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G4R_MAX_LAYERS 8

/* gru4rec.py:136-143 (set_loss_function) */
enum { G4R_LOSS_XE = 0, G4R_LOSS_BPR_MAX = 1, G4R_LOSS_TOP1_MAX = 2, G4R_LOSS_BPR = 3, G4R_LOSS_TOP1 = 4,
       G4R_LOSS_XE_LOGIT = 5 };
/* gru4rec.py:144-161 (set_final_activation / set_hidden_activation) */
enum { G4R_ACT_LINEAR = 0, G4R_ACT_RELU = 1, G4R_ACT_TANH = 2, G4R_ACT_LEAKY = 3, G4R_ACT_ELU = 4,
       G4R_ACT_SELU = 5, G4R_ACT_SOFTMAX = 6, G4R_ACT_SOFTMAX_LOGIT = 7 /* final activation only; plain softmax when
       predicting, gru4rec.py:490-491,499-500 */ };
/* gru4rec.py:438-470: where the GRU input comes from */
enum { G4R_EMBED_CONSTRAINED = 0 /* Wy shared, :438-448 */, G4R_EMBED_SEPARATE = 1 /* E, :449-456 */,
       G4R_EMBED_ONEHOT = 2 /* no embedding, layer 0 reads rows of Wx[0] (I x 3D), :457-470; the constructor default */ };
/* gru4rec.py:392-399,411-418: learning-rate adaptation (`adapt`); NONE = plain SGD */
enum { G4R_ADAPT_ADAGRAD = 0, G4R_ADAPT_RMSPROP = 1, G4R_ADAPT_ADADELTA = 2, G4R_ADAPT_ADAM = 3, G4R_ADAPT_NONE = 4 };
/* evaluation.py:62-65 */
enum { G4R_RANK_STANDARD = 0, G4R_RANK_CONSERVATIVE = 1, G4R_RANK_MEDIAN = 2,
       G4R_RANK_TIEBREAKING = 3 /* :55: scores + uniform * 1e-10 in fp32 (Philox stream keyed by the model seed, the evaluation step,
       row and candidate column instead of the reference's MRG stream), then ranked as STANDARD */ };

/* Constructor arguments of the reference's GRU4Rec (gru4rec.py:97-135) that the hot path needs. */
typedef struct g4r_config {
    int32_t n_items;
    int32_t n_layers;
    int32_t layers[G4R_MAX_LAYERS];
    int32_t batch_size;
    int32_t n_sample;            /* additional negatives per step (0 = in-batch only) */
    int32_t loss;                /* G4R_LOSS_* */
    int32_t final_act;           /* G4R_ACT_* */
    float   final_act_p0, final_act_p1;
    int32_t hidden_act;          /* G4R_ACT_* (not softmax) */
    float   hidden_act_p0, hidden_act_p1;
    int32_t embed_mode;          /* G4R_EMBED_* */
    int32_t embedding;           /* width of E when embed_mode == SEPARATE */
    float   learning_rate, momentum, lmbd, bpreg, logq, sample_alpha;
    float   dropout_p_hidden, dropout_p_embed;
    int64_t sample_store;        /* ints in the negative-sample store (gru4rec.py:515,547) */
    uint64_t seed;               /* Philox key for sampler + dropout */
    int32_t device;              /* HIP device ordinal */
    int32_t rank, nranks;        /* data-parallel rank layout (1 process per GPU) */
    int32_t use_graph;           /* capture steady-state steps into a hipGraph */
    float   smoothing;           /* label smoothing of cross-entropy / xe_logit, gru4rec.py:226-235 */
    int32_t adapt;               /* G4R_ADAPT_* */
    float   adapt_p0, adapt_p1;  /* adapt_params (rmsprop / adadelta: decay; adam: beta1, beta2), gru4rec.py:301-304,342,368 */
    float   grad_cap;            /* global gradient-norm clip, 0 = off, gru4rec.py:386-389 */
    int32_t sparse_exact;        /* N > 1 only.  0: item rows stay GPU-local between reconciliations (north_star; g4r_comm_sync_sparse /
                                    g4r_set_sync_every).  1: EXACT replicas -- every step each rank's per-occurrence gradient rows of the
                                    gathered item rows are all-gathered and every rank applies all of them in rank order, with the
                                    reference's duplicate semantics over the concatenated occurrence list (gru4rec.py:335-340,407-431):
                                    replicas never diverge, no base copies, no sync_every (SURVEY 8e option 3).  Measured with virtual
                                    ranks: applying ALL ranks' per-occurrence Adagrad steps diverges from four
                                    ranks on, like summed deltas did in round 3 -- the popular items receive N full-size steps.
                                    2: the same exchange, MEAN form: an item's parameter increment is the mean over the ranks that
                                    touch it of each rank's own increment, its Adagrad accumulator the sum of those ranks'
                                    last-occurrence increments -- the GPU-local mode's reconciliation rule taken every step.
                                    3 (what GRU4Rec.sparse_exact = True selects): REDUCE form.  All ranks draw the same negatives; the
                                    gradient rows of those shared columns are summed over the ranks, the ranks' input / target
                                    occurrences are listed one rank behind the other, all scaled by 1 / nranks: the occurrence list of
                                    ONE batch of nranks x batch_size rows sharing one row of negatives (gru4rec.py:436-437), updated
                                    with the reference's rule.  Modes 1-3 want the same `seed` on every rank (one sample stream; dropout
                                    masks are keyed by seed + 7919 * rank inside the library); the ranks' raw dense gradients travel in
                                    the same all-gathered block and every rank sums them in rank order: ONE collective per step.
                                    Mode 3 checks every step that the ranks' negatives are the same ids (a rank in the padded tail of
                                    its plan holds none: the ids then come from the first rank that has them); a mismatch turns the
                                    step's cost into NaN on every rank -- the run stops at the caller's NaN check */
    int32_t defer_updates;       /* 1 (or G4R_DEFER=1 in the environment): single GPU, Adagrad without momentum / lmbd, graph replay -- the update of a
                                    gathered row (gru4rec.py:420-431) whose item is not gathered again inside the current window of 16 steps (known
                                    ahead from the plan and the sample store) is applied by ONE flush launch at the end of the window instead of by
                                    its step's update launch: the same operands and arithmetic, bit-identical results, a launch long enough for the
                                    HBM (~60 % of 8 TB/s at BASELINE configs[2]); costs 2-5 % of the step rate */
} g4r_config;

typedef struct g4r_model g4r_model;

/* ---- lifetime ------------------------------------------------------------------------------- */
int  g4r_device_count(void);
const char* g4r_last_error(void);
const char* g4r_version(void);
int  g4r_sizeof_config(void);   /* sizeof(g4r_config), for binding self-checks */
/* replaces GRU4Rec.init()'s theano.shared allocations, gru4rec.py:267-294 */
int  g4r_create(const g4r_config* cfg, g4r_model** out);
void g4r_destroy(g4r_model* m);

/* ---- parameters (gru4rec.py:742-781 savemodel/loadmodel get_value/set_value) ------------------ */
/* name: "Wy" [I,D], "By" [I], "E" [I,emb], "Wx" [in,3D], "Wh" [D,D], "Wrz" [D,2D], "Bh" [3D], "H" [B,D];
 * optimizer state with prefix "acc_" / "vel_" (gru4rec.py:331,401,425).  `layer` ignored for Wy/By/E. */
int g4r_set_param(g4r_model* m, const char* name, int32_t layer, const float* host, int64_t count);
int g4r_get_param(g4r_model* m, const char* name, int32_t layer, float* host, int64_t count);

/* ---- negative sampling (gru4rec.py:539-571; kernel semantics custom_theano_ops.py:318-349) ---- */
/* cum_p: float32 cumulative supp**alpha (P, :543-545,556); lq_tgt/lq_smp: logQ tables (:495), may be NULL */
int g4r_set_popularity(g4r_model* m, const float* cum_p, const float* lq_tgt, const float* lq_smp, int64_t n);
/* test hook: overwrite the device sample store (rows x n_sample) and freeze refills */
int g4r_set_sample_store(g4r_model* m, const int32_t* store, int64_t rows);
int g4r_get_sample_store(g4r_model* m, int32_t* store, int64_t rows);
int64_t g4r_sample_store_rows(g4r_model* m);

/* ---- epoch plan: the (X, Y, M, R) argument stream of train_function, gru4rec.py:599-651 ------- */
/* Host-side scheduler (no GPU needed): restates the while/for loop of fit() for a whole epoch.
 * Pass NULL outputs to only count.  Returns the number of steps T (or <0); *n_compact receives the
 * number of batch-shrink events (gru4rec.py:647-651).  compact_maps is [n_compact][batch_size]:
 * new row j takes old row map[j] (-1 = none). */
int64_t g4r_build_plan(const int32_t* offset_sessions, int64_t n_sessions, const int64_t* session_order,
                       const int32_t* data_items, int32_t batch_size, int32_t n_sample,
                       int32_t* in_idx, int32_t* out_idx, uint8_t* reset, int32_t* M,
                       int64_t* compact_steps, int32_t* compact_maps, int64_t max_steps, int64_t max_compact,
                       int64_t* n_compact);
/* upload a plan: in_idx/out_idx [T,B] int32, reset [T,B] uint8, M [T].  Also captures and instantiates the step graph
 * (nothing executes), so that the first g4r_train_steps call does not pay for it. */
int g4r_set_plan(g4r_model* m, const int32_t* in_idx, const int32_t* out_idx, const uint8_t* reset,
                 const int32_t* M, int64_t T, const int64_t* compact_steps, const int32_t* compact_maps,
                 int64_t n_compact);

/* ---- the hot path: replaces `train_function(in_idx, y, len(iters), reset)`, gru4rec.py:584,623 - */
/* runs plan steps [t0, t0+n): gather -> GRU -> sampled scoring -> loss -> backward -> Adagrad, with no
 * host round trip in between; per-step costs stay on the device until g4r_get_losses. */
int g4r_train_steps(g4r_model* m, int64_t t0, int64_t n_steps);
int g4r_get_losses(g4r_model* m, int64_t t0, int64_t n, float* out);     /* cost of :623 per step */
int g4r_synchronize(g4r_model* m);
int64_t g4r_global_step(g4r_model* m);
/* checkpoint resume ("optimizer-state save"): the number of sample-store refills so far, and the setter that
 * puts a fresh handle where a saved run stopped (global step: dropout counters, store row pointer; refills: store contents) */
int64_t g4r_refills(g4r_model* m);
int g4r_set_step_counters(g4r_model* m, int64_t global_step, int64_t refills);
/* average HIP-event time (ms) per launch of each step kernel since the last reset; names via index.
 * g4r_profile: 0 = off, 1 = on (steps are launched eagerly with start / stop events attached to every dispatch), 2 = on, with the
 * two roles of the single-GPU update launch as launches of their own (k_dense_grad, k_sparse_update): the embedding
 * gather / scatter north_star prices against the HBM roofline, timed ALONE (results are identical either way). */
int g4r_kernel_time(g4r_model* m, int32_t which, const char** name, double* total_ms, int64_t* launches);
int g4r_profile(g4r_model* m, int32_t enable);

/* hidden state (gru4rec.py:589-590 zeroing, :647-651 compaction; evaluation.py:134-139) */
int g4r_reset_hidden(g4r_model* m);

/* ---- prediction: replaces the compiled `self.predict` / evaluate functions --------------------- */
/* gru4rec.py:691-711 (+ symbolic_predict :729-741): allocate prediction state for `batch` rows */
int g4r_predict_begin(g4r_model* m, int32_t batch);
/* zero rows where zero_mask[0..n_mask) != 0 (n_mask <= batch of g4r_predict_begin; later rows keep their state), then keep
 * rows in `keep_rows` order (NULL = identity), evaluation.py:134-139, gru4rec.py:712-717 */
int g4r_predict_hidden(g4r_model* m, const uint8_t* zero_mask, int32_t n_mask, const int32_t* keep_rows, int32_t n_keep);
/* forward only; scores[m, n_sel] = final_act(h Wy[item_idx]^T + By) (all items if item_idx NULL); out may be NULL */
int g4r_predict_step(g4r_model* m, const int32_t* in_idx, int32_t mrows, const int32_t* item_idx, int64_t n_sel,
                     float* out_scores);
/* rank of column target_col[i] of row i among columns [col_begin, n_sel) of the last g4r_predict_step's
 * scores (evaluation.py:56-65; col_begin = 0 for the all-items case, = M when `items` were given) */
int g4r_rank_targets(g4r_model* m, const int32_t* target_col, int32_t mrows, int64_t col_begin, int32_t mode,
                     float* ranks);

/* The whole of evaluation.evaluate_gpu (evaluation.py:86-147) as ONE call with no host round trip per step: the
 * session-parallel test loop comes as a plan (g4r_build_plan on the test sessions in id order with n_sample = 1: the loop of
 * evaluation.py:96-139 is the loop of fit), every step runs the GRU forward, scores all items (items == NULL) or
 * [targets | items], ranks the targets (mode = G4R_RANK_*), and adds #(rank <= cut) and sum 1/rank for each cut-off into
 * device accumulators; hidden rows of finished sessions are zeroed / dropped on the device (evaluation.py:134-139).
 * Outputs: recall_sum[n_cut], mrr_sum[n_cut] (divide by *n_events). */
int g4r_evaluate(g4r_model* m, const int32_t* in_idx, const int32_t* out_idx, const uint8_t* reset, const int32_t* M, int64_t T,
                 int32_t batch, const int64_t* compact_steps, const int32_t* compact_maps, int64_t n_compact,
                 const int32_t* items, int64_t n_items_sel, const int32_t* cutoffs, int32_t n_cut, int32_t mode,
                 double* recall_sum, double* mrr_sum, int64_t* n_events);

/* ---- input pipeline: the pandas work in front of fit() for TAB separated files -------------
 * run.py:45-78 `pd.read_csv(sep='\t', usecols, dtype={session: int32, item: str})` + gru4rec.py:534-538
 * `itemids = data[item_key].unique(); itemidmap = Series(arange, index=itemids); data['ItemIdx'] = itemidmap[...]`.
 * Multi-threaded mmap parse; item indices are assigned in order of first appearance (= pandas unique()).  Host only.
 * Returns 0, G4R_IO_UNSUPPORTED when the file needs a general CSV parser (quoted fields, missing values, session ids
 * that are not int32: the caller falls back to pandas), or <0 with g4r_last_error().  time_col may be NULL.
 * n_threads: 0 = all cores (at most 64, at most one per MiB), > 0 = at most that many, < 0 = exactly -n_threads (tests). */
#define G4R_IO_UNSUPPORTED 1
typedef struct g4r_events g4r_events;
int g4r_events_load(const char* path, const char* session_col, const char* item_col, const char* time_col, int32_t n_threads,
                    g4r_events** out);
int64_t g4r_events_rows(const g4r_events* ev);
int64_t g4r_events_items(const g4r_events* ev);
int64_t g4r_events_item_bytes(const g4r_events* ev);     /* total length of the distinct item-id strings */
int32_t g4r_events_time_kind(const g4r_events* ev);      /* 0 = no time column, 1 = int64, 2 = float64 */
/* session[n_rows] int32, item_idx[n_rows] int32, time[n_rows] (int64 or double by time_kind), item_off[n_items + 1] offsets
 * into item_bytes (item ids in index order, not terminated).  Any output may be NULL. */
int g4r_events_copy(const g4r_events* ev, int32_t* session, int32_t* item_idx, void* time, int64_t* item_off, char* item_bytes);
void g4r_events_free(g4r_events* ev);

/* ---- multi-GPU (new: the reference is single-GPU).  RCCL all-reduce of dense GRU gradients ---- */
int g4r_comm_unique_id(char* out128);                         /* rank 0: ncclGetUniqueId */
int g4r_comm_init(g4r_model* m, const char* id128, int32_t nranks, int32_t rank);
/* Reconciliation of the GPU-local item tables (Wy / By / E and their optimizer state; north_star: "sparse embedding rows stay
 * GPU-local").  For every row some rank rewrote since the last call, every replica ends at
 *     base + sum over ranks q, in rank order, of (value on rank q - base),      base = the common value at the last call,
 * so each rank's updates are kept in full (a row one rank alone trained ends at that rank's value up to fp32 rounding) and the
 * replicas are bit-identical afterwards.  Packed (id list, delta rows) parts are
 * all-gathered in item-id ranges: the traffic follows the number of touched rows.  Called by fit() at the end of an epoch. */
int g4r_comm_sync_sparse(g4r_model* m);
/* combine rule of the reconciliation, per kind of plane: parameters (Wy / By / E and their velocities) and optimizer statistics
 * (Adagrad / RMSprop accumulators, Adam moments and counters).  G4R_SYNC_SUM: base + sum of the deltas of the ranks that touched
 * the row; G4R_SYNC_MEAN: base + their mean.  Default: parameters MEAN (N full-size steps from one starting point must not add up:
 * measured); statistics SUM with adapt = adagrad (squared gradients of all ranks' events add up as they would
 * on one GPU) and MEAN with rmsprop / adadelta / adam, whose statistics are moving averages: summed deltas of a decayed
 * average go negative once several ranks touch a row (NaN in the next sqrt) -- tests/test_gpu_virtual_ranks.py. */
#define G4R_SYNC_SUM 0
#define G4R_SYNC_MEAN 1
int g4r_sync_set_rule(g4r_model* m, int32_t param_rule, int32_t stat_rule);
/* allocates the touched-row bitmap and the base copies, base = the tables as they are now (g4r_comm_init calls it;
 * g4r_set_param of an item table afterwards also sets its base) */
int g4r_sync_enable(g4r_model* m);
/* The two halves of g4r_comm_sync_sparse without RCCL, for tests that emulate ranks with several handles on one GPU.
 * group 0: Wy / By rows, 1: E rows.  export: sorted ids of the touched rows and, plane after plane (Wy, acc_Wy, [vel ...], By,
 * acc_By, ...), their delta rows [n][W]; returns n (pass NULL pointers to size the buffers; g4r_sync_row_floats = sum of the plane
 * widths).  import: the parts of ALL ranks in rank order (this handle's own included) -> reset, add, new base, bitmap cleared. */
int64_t g4r_sync_row_floats(g4r_model* m, int32_t group);
int64_t g4r_sync_export(g4r_model* m, int32_t group, int32_t* ids, float* rows, int64_t cap_rows);
int g4r_sync_import(g4r_model* m, int32_t group, int32_t nparts, const int64_t* counts, const int32_t* const* ids,
                    const float* const* rows);
/* Virtual ranks: `n` handles on ONE device stand in for the n ranks of a data-parallel run (handle q created with rank = q,
 * nranks = n, no communicator; each with its own plan of the same length).  Runs plan steps [t0, t0 + n_steps) in lock-step:
 * per step every handle's kernels up to the dense gradients, the n gradient buffers summed in rank order (what the RCCL
 * all-reduce of the real run delivers), every handle's dense apply (divides by nranks) and GPU-local sparse update.  Item tables
 * are reconciled by the caller (g4r_sync_export / g4r_sync_import).  For validating the N > 1 training semantics -- e.g.
 * Recall@20 of 2 / 8 ranks against 1 (evaluation.py:62-75) -- on a one-GPU box; not a fast path. */
int g4r_virtual_train_steps(g4r_model* const* ms, int32_t n, int64_t t0, int64_t n_steps);
/* the dense form of g4r_comm_sync_sparse (item tables of up to G4R_SYNC_DENSE_MB = 64 MB per group: [n_items][plane widths + 1]
 * delta buffers packed, summed and applied on the device) with the sum taken in process over n handles of one device */
int g4r_virtual_sync_dense(g4r_model* const* ms, int32_t n);
/* k > 0: g4r_train_steps reconciles the item tables itself every k steps (counted across calls), between two steps and without
 * leaving the stream -- where every table takes the dense form and a communicator exists: returns 1; returns 0 when the caller has to
 * call g4r_comm_sync_sparse itself (tables too large for the dense form); k = 0 switches it off. */
int g4r_set_sync_every(g4r_model* m, int32_t k);
int g4r_comm_min_i64(g4r_model* m, int64_t* value);            /* in-place min over ranks */
int g4r_comm_max_i64(g4r_model* m, int64_t* value);            /* in-place max over ranks: the common plan length (shorter plans are
                                                                  padded with M = 0 steps so that every rank issues the same all-reduces) */
int g4r_comm_nranks(g4r_model* m);                             /* ranks RCCL reports for the communicator (1 without one) */
/* One-shot all-reduce of the dense GRU gradients through peer memory -- the switch next to the RCCL all-reduce (new: the reference
 * is single-GPU; north_star asks for the gradients of gru4rec.py:383-384's dense parameters to be all-reduced every step).  The
 * gradients are a few hundred KB, so instead of a ring every rank reads the other ranks' buffers over its point-to-point xGMI
 * links and adds them in rank order itself (k_p2p_allreduce: per-workgroup stamped hand-offs, two alternating buffers, no
 * host work per step, captured in the step graph).  One node, at most 8 ranks.
 *   g4r_p2p_enable   on a handle with a communicator: allocates this rank's exchange region, all-gathers the 64-byte IPC handles
 *                    through RCCL, maps the peers.  Collective.  GRU4Rec does this when G4R_P2P=1.
 *   g4r_p2p_export / g4r_p2p_attach   the same with the handles carried by the caller (handles: nranks x 64 bytes in rank order);
 *                    needs no RCCL, so two PROCESSES on one device can be each other's ranks (tests/test_gpu_p2p.py).
 * A peer that does not publish within G4R_P2P_TIMEOUT_MS (default 20000) makes g4r_train_steps fail instead of hang. */
int g4r_p2p_enable(g4r_model* m);
int g4r_p2p_export(g4r_model* m, char* out_handle64);
int g4r_p2p_attach(g4r_model* m, const char* handles, int32_t nranks, int32_t rank);
int g4r_p2p_active(g4r_model* m);                              /* 1 when the step's all-reduce is the peer-memory one */

/* ---- debugging / tests ---------------------------------------------------------------------- */
/* copy a named intermediate of the most recent step (e.g. "scores", "dS", "dV0", "hd0") */
int g4r_get_debug(g4r_model* m, const char* name, float* host, int64_t count);
int g4r_selftest_mfma(float* max_abs_err);
/* Row gather / scatter micro-benchmark on a table of n_items x W floats (fresh allocation, random rows, every launch its own
 * rows): mode 0 gather to a compact buffer, 1 gather consumed in registers (what the step's fused gathers do), 2 Adagrad scatter
 * (gradient row read + parameter r/w + accumulator r/w).  Replaces, as an object of measurement, the reference's gather kernel
 * custom_theano_ops.py:505-519 and its sparse Adagrad scatter gru4rec.py:335-340,428-431.  kernel_us: mean dispatch-to-completion
 * time of a launch (HIP events attached to the dispatch); wall_us: back-to-back launches on one stream, per launch. */
int g4r_bench_rows(int32_t device, int64_t n_items, int32_t W, int64_t rows_per_launch, int32_t launches, int32_t mode, uint64_t seed,
                   double* kernel_us, double* wall_us);                    /* 16x16x4 f32 MFMA layout check */
/* Test support (tests/test_gpu_stress.py): `launches` passes of a streaming read-modify-write over `mbytes` MiB of device memory
 * on a stream of their own, queued asynchronously -- HBM / Infinity-Cache load next to the caller's training steps, the condition
 * under which round 3's stale-register pipeline (mutation build 4) went wrong and an idle GPU never shows.  g4r_stress_stop waits
 * for the passes and frees the buffer.  Nothing of the reference corresponds to it. */
int g4r_stress_start(int32_t device, int64_t mbytes, int32_t launches, void** handle);
int g4r_stress_stop(void* handle);

#ifdef __cplusplus
}
#endif
#endif
