// g4r_lean_kernels.cuh -- part of g4r_step_kernels.cuh (included there, in order; needs its prelude).  Holds the GRU forward and backward of a
// NARROW layer (in, D <= 128: BASELINE configs[0], [1], [4]) as four latency-lean launches: k_gru_v, k_gru_h (forward), k_gru_da, k_gru_dy (backward).
//
// Why four launches where round 2 had fused two (k_gru_fwd_fused / k_gru_bwd_fused).  Round 6 measured the pieces (tools/probes/chain_probe.hip,
// profiles/r06_chain_probe.txt): a dependent launch in the step's hipGraph costs 1.5 us, a dependent memory round trip inside a kernel 0.15-0.45 us,
// a workgroup pulls 144 KB of freshly rewritten weights in 1.2 us, eight dependent scalar loads cost 0.1 us -- but an in-launch hand-off between
// workgroups costs 1.3-1.7 us, and ONE wave issues one instruction per ~7 clocks when each depends on the last (512 dependent v_fma: 1.2 us).
// The fused kernels pay for both: the candidate needs (H r) Wh over ALL columns of r, so each of the four column tiles of a row block recomputes
// r (350 of its 550 fp32 MFMAs, 80 of its 147 KB of weights) on the 32 CUs the launch occupies -- 2.4 us of MFMA issue + 2.6 us of operand requests
// + 2.3 us of LDS epilogues + four barriers in a 8.8 us body (tools/clk.py, profiles/r06_clk_cfg2_baseline.txt).  A kernel boundary IS the cheap
// cross-workgroup exchange on this chip.  So: cut at the two data dependencies (r before the candidate; dr' before dy), and make each piece as
// short as the hardware allows -- which at these sizes means as FEW INSTRUCTIONS PER WAVE as possible:
//   * one 16 x 16 output tile per workgroup (56-168 workgroups per launch instead of 32), its K range dealt out one 16-deep super-step per wave;
//   * NO LDS staging of operands: every lane loads its MFMA fragments straight from global memory.  A[row][k] rows are read as float4 along k --
//     lane (li, lg) of super-step s holds quad 4 s + lg, and MFMA u of the super-step multiplies component u: a permutation of k inside 16 that both
//     operands share -- B[k][n] as the matching four dwords (row-major weights) or as float4 along k (transposed weights: the backward);
//   * everything a kernel reads from the model comes out of ONE compact argument block (LeanV / LeanH / LeanDa / LeanDy, device resident, built at
//     g4r_create from pointers that never change afterwards): two or three wide scalar loads at the top instead of a dozen lazy descriptor reads;
//     uniform bases + 32-bit lane offsets (global_load ..., v_off, s[base]), a 3-D grid instead of integer divisions;
//   * every load of a wave is issued before the first is consumed (one memory round trip per wave, two where an index comes first);
//   * one barrier per kernel: the waves' K-slice partial tiles go to LDS component-wise, waves 0..3 each add one component (wave order: fixed)
//     and run the epilogue for one output value per lane.
// The r-dependency of the backward is turned into a sum the consumer does anyway: k_gru_da owns 16 COLUMNS of da (K slice j of dr' = da Wh^T) and
// leaves dr'_j = da[:, slice j] Wh[:, slice j]^T for all D columns as a partial plane; k_gru_dy adds the <= 8 planes in slice order while it builds
// its A fragments ((sum) * H * r (1 - r): the factor is applied after the sum, so the partial planes are plain linear pieces).
// Arithmetic: the same fp32 MFMA fmaf chains, in a different (fixed) order than the fused kernels -- results agree with the oracle within the
// same bounds (tests/test_gpu_parity.py::test_first_step_intermediates and the goldens run both forms), graph replay == eager bit for bit.
// Math: gru4rec.py:471-479 (forward), T.grad :383-384 (backward); SURVEY section 8 a6 / a9.
#pragma once

#define LN_MAXD 128      // in, D <= 128: at most 8 super-steps of 16 per K segment

// loads through a uniform base + 32-bit byte offset (the saddr form: no 64-bit lane arithmetic)
__device__ __forceinline__ float ldu(const GAS float* base, unsigned boff) { return *(const GAS float*)((const GAS char*)base + boff); }
__device__ __forceinline__ int ldu_i(const GAS int* base, unsigned boff) { return *(const GAS int*)((const GAS char*)base + boff); }
__device__ __forceinline__ float4 ldu4(const GAS float* base, unsigned boff) { return *(const GAS float4*)((const GAS char*)base + boff); }
__device__ __forceinline__ void stu(GAS float* base, unsigned boff, float v) { *(GAS float*)((GAS char*)base + boff) = v; }
__device__ __forceinline__ void stu4(GAS float* base, unsigned boff, float4 v) { *(GAS float4*)((GAS char*)base + boff) = v; }

// phase stamps of workgroup (1, 1), thread 0 (G4R_CLK_TRACE builds; tools/clk_lean.py): slot base per kernel, 100 MHz wall clock
#if defined(G4R_CLK_TRACE)
#define LCLK_INIT(dbgp, slot) GAS long long* clk_ = ((dbgp) && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && threadIdx.x == 0) ? (dbgp) + (slot) : nullptr
#define LCLK(i) do { if (clk_) clk_[i] = wall_clock64(); } while (0)
#define LCLK_USE(x) do { if (clk_ && (x) == 123.456f) clk_[15] = 0; } while (0)
// every workgroup's (first stamp, last stamp) -> dbgtile[base + linear workgroup id] (tools/clk_lean.py: the launch's span on the wall clock)
#define LSPAN_BEGIN(tilep, base) GAS long long* span_ = ((tilep) && threadIdx.x == 0) ? (tilep) + 8 * (size_t)((base) + (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) : nullptr; \
    if (span_) span_[0] = wall_clock64()
#define LSPAN_END() do { if (span_) span_[1] = wall_clock64(); } while (0)
#else
#define LCLK_INIT(dbgp, slot)
#define LCLK(i)
#define LCLK_USE(x)
#define LSPAN_BEGIN(tilep, base)
#define LSPAN_END()
#endif

// argument blocks (device resident, one per layer; g4r_host_create.hpp: build_lean_args).  Pointers only to buffers that live as long as the model.
struct LeanV {
    GP(const float) Wx; GP(const float) Wrz; GP(const float) Bh; GP(const float) H0; GP(const float) H1;
    GP(const float) ysrc;        // layer 0: the input table (Wy / E); else the lower layer's output hd[l - 1]
    GP(const int) cur_in;        // staged in_idx row of the step (layer 0)
    GP(float) Vc; GP(float) r; GP(float) Hr; GP(float) z; GP(float) yin0;
    GP(int) occ_idx; GP(int) occ_fl;      // occ_fl: already at the input table's block
    GP(StepState) st;
    unsigned long long seed;
    int B, D, IN, R, first, pub_fl;
    float drop_e;
    int n_items;
    GP(long long) dbg; GP(long long) dbgtile;
};
struct LeanH {
    GP(const float) Wh; GP(const float) H0; GP(const float) H1; GP(const float) Hr; GP(const float) Vc; GP(const float) z;
    GP(const int) cur_rst;       // staged reset flags of the step (one int per row)
    GP(float) c; GP(float) hd;
    GP(const StepState) st;
    unsigned long long seed;
    int B, D, hidden_act, stream;
    float ha_p0, ha_p1, drop_h, pad;
    GP(long long) dbg; GP(long long) dbgtile;           // phase stamps (G4R_CLK builds)
};
struct LeanDa {
    GP(const float) Wh; GP(const float) H0; GP(const float) H1; GP(const float) z; GP(const float) c;
    GP(const float) dsrc;        // slabs of dh (top layer) / K-slice partial sums of a wide upper layer's dy / the upper layer's dy
    GP(float) dV; GP(float) drp;
    GP(const StepState) st;
    unsigned long long seed;
    int B, D, ks, hidden_act, stream, pad0;      // ks: planes of dsrc to add (1: dsrc is dh itself)
    float ha_p0, ha_p1, drop_h, pad;
    GP(long long) dbg; GP(long long) dbgtile;
};
struct LeanDy {
    GP(const float) Wx; GP(const float) H0; GP(const float) H1; GP(const float) r; GP(const float) drp;
    GP(float) dV; GP(const int) occ_idx; GP(const int) occ_fl; GP(float) accT;
    GP(float) dSx; GP(float) dAx; GP(float) dylo;
    GP(const StepState) st;
    unsigned long long seed;
    long long dSx_stride;
    int B, D, IN, layer0, generic, defer_mask;
    float lr, drop_e;
    GP(long long) dbg; GP(long long) dbgtile;
    int n_items, pad;
};

struct LeanS {
    GP(int) col_item; GP(int) occ_idx; GP(int) occ_fl;      // occ_idx: at the Y | samples part (offset B); occ_fl: the Wy / By table's block
    const DevModel* mp;                                         // (the logQ tables are set after g4r_create: read through the descriptor)
    GP(long long) dbg; GP(long long) dbgtile;
    int R, pub_fl;
    float logq, pad;
};

struct LeanB {
    GP(float) accBy; GP(const int) occ_fl; GP(float) dSy; GP(float) dAy; GP(float) dSBy; GP(float) dABy; GP(float) dhpart;
    GP(long long) dbg; GP(long long) dbgtile;
    long long dSy_stride, dSBy_stride;
    int defer_mask, generic, nA, ndh, nrb, ndb;      // nA: role A workgroups; ndh: 64-wide d blocks of role A (D + 1 columns); role B: nrb row blocks x ndb d blocks per slab
    float lr, pad;
};

struct LeanU {
    const DevModel* mp; StepState* st;      // the bookkeeping workgroup works through the descriptor
    GP(float) Wy; GP(float) E; GP(float) accWy; GP(float) accE; GP(float) velWy; GP(float) velE;
    GP(float) By; GP(float) accBy; GP(float) velBy;
    GP(const float) dAx; GP(const float) dAy; GP(const float) dABy;
    GP(float) dense_p; GP(float) dense_acc; GP(float) dense_vel; GP(const float) yin0;
    GP(const int) meta;
    GP(long long) dbg; GP(long long) dbgtile;
    int n_items, constrained, wE, wY;
    float lr, mom, lmbd, pad;
};

// ---------------------------------------------------------------------------------------------
// How every kernel below is laid out in time (tools/clk_lean.py, profiles/r06_clk_lean_*.txt): a line the previous launch wrote costs
// 0.4-0.6 us to fetch, the step state (rewritten every step) as much -- so NO load may wait for the state or for the argument block:
//   t = 0     the pointers and sizes the first loads need arrive in SGPRs with the wave (kernarg preload: 16 dwords); the step's (g, M, t)
//             -- staged behind cur_in by the previous step's bookkeeping -- row items, weights and activations are requested at once as
//             VECTOR loads, rows clamped by the batch size (not by the step's M), BOTH H buffers.  (hipcc sinks a scalar load of the state
//             to its first use, behind the branches in front of it: its latency was fully exposed; a vector load stays where it is written.)
//   ~0.2 us   the argument block is there (lean_pin: every field in SGPRs at one point, instead of a lazy scalar load per use);
//   ~0.5 us   state and first-level loads are there: masks (row < M), H parity, second-level gathers (item -> table row);
//   then      MFMAs, one LDS hand-over, epilogue.
// A CU's vector memory path moves 64 B per clock (144 KB = 1.2 us, chain_probe): waves of one workgroup must not load the same bytes
// eight times (k_gru_da's first form: 106 KB per workgroup = 0.8 us).
template <typename T> __device__ __forceinline__ void lean_pin1(T v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(v));
#endif
}
template <typename... T> __device__ __forceinline__ void lean_pin(T... v) { (lean_pin1(v), ...); }
// A value that must be COMPLETE here (not recomputed inside a later branch).  On gfx9 / CDNA stores count against vmcnt like loads: when
// hipcc sinks the last use of a loaded operand into each of several conditional store blocks, every block gets its own
// `s_waitcnt vmcnt(0)` -- which, from the second block on, waits for the previous block's STORE to be acknowledged: the stores of an
// epilogue then go out one memory round trip apart (4 conditional row stores: +1.5 us; found in round 6, profiles/r06_experiments.md).
// Computing the stored values first and pinning them keeps the blocks free of waits.
__device__ __forceinline__ void lean_keep(float& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}
struct LeanState { unsigned g; int M; unsigned t_lo; };
// staged (g lo, g hi, M, t lo, t hi) -> wave-uniform registers (every lane loaded the same 16 bytes)
__device__ __forceinline__ LeanState lean_state(int4 mt) {
    LeanState s;
    s.g = (unsigned)__builtin_amdgcn_readfirstlane(mt.x); s.M = __builtin_amdgcn_readfirstlane(mt.z); s.t_lo = (unsigned)__builtin_amdgcn_readfirstlane(mt.w);
    return s;
}

// Forward, launch 1: V = [y | H] [Wx ; 0 | Wrz] + Bh for ONE 16 x 16 tile per workgroup (gru4rec.py:472-473).
// grid (ceil(D / 16), ceil(B / 16), 3): blockIdx.z = part; part 0: candidate input part V_c = y Wx[:, 0:D] (K = in) -> Vc; part 1: r = sigmoid(.),
// Hr = H r; part 2: z = sigmoid(.)  (K = in + D).  Eight waves: wave w takes super-step w of the y segment AND of the H segment.  Layer 0 (L0)
// gathers Wy[X] / E[X] rows (+ embedding dropout, DROPE); workgroup (0, row block, 0) publishes them (yin0: the dense-gradient tiles read them
// back) and the X part of occ_idx / occ_fl; workgroup (0, 0, 0) republishes the step state for the kernels that read StepState::*_b.
template <bool L0, bool DROPE>
__global__ __launch_bounds__(512) void k_gru_v(const LeanV* __restrict__ ap, StepState* st_, const int* cur_in_, const float* Wx_, const float* Wrz_,
                                               const float* H0_, const float* H1_, unsigned dims, unsigned B) {
    __shared__ float sJ[4 * 8 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned part = blockIdx.z, m0 = blockIdx.y * 16, rowA = m0 + li;
    const unsigned D = dims & 0xFFFFu, IN = dims >> 16;
    const GAS float *Wx = (const GAS float*)Wx_, *Wrz = (const GAS float*)Wrz_, *H0 = (const GAS float*)H0_, *H1 = (const GAS float*)H1_;
    const GAS int* cur_in = (const GAS int*)cur_in_;
    // row item first (the gather waits for it), then the step's (g, M, t); both staged by the previous step's bookkeeping
    const unsigned rowB = min(rowA, B - 1);
    int item = 0;
    if (L0) item = ldu_i(cur_in, 4u * rowB);
    const int4 mt = ldi4(cur_in + 2 * B);
    const unsigned ncol = 16 * blockIdx.x + li, nc = min(ncol, D - 1);
    const unsigned Q = 4 * wid + lg;                   // quad of this lane in either segment
    const bool oky = Q < (IN >> 2), okh = part != 0 && Q < (D >> 2);
    const unsigned Qy = oky ? Q : 0, Qh = okh ? Q : 0;
    // B fragments: Wx[4 Qy + u][part D + n] (row stride 3 D), Wrz[4 Qh + u][(part - 1) D + n] (row stride 2 D)
    const unsigned D3b = 12 * D, D2b = 8 * D;
    const unsigned offx = 4 * Qy * D3b + 4 * (part * D + nc), offh = 4 * Qh * D2b + 4 * ((part ? part - 1 : 0) * D + nc);
    float bx[4], bh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bx[u] = ldu(Wx, offx + u * D3b);
#pragma unroll
    for (int u = 0; u < 4; ++u) bh[u] = ldu(Wrz, offh + u * D2b);
    // the H quad of the lane's row and the epilogue's H element, out of both buffers (the parity comes with the state)
    const unsigned offa = 4 * (rowB * D + 4 * Qh);
    const float4 ah0 = ldu4(H0, offa), ah1 = ldu4(H1, offa);
    const unsigned rowe = m0 + 4 * lg + (wid & 3);     // epilogue of waves 0 .. 3: component rg = wave
    const unsigned offe = 4 * (min(rowe, B - 1) * D + nc);
    const float he0 = ldu(H0, offe), he1 = ldu(H1, offe);
    // ---- the argument block is needed from here
    const LeanV a = *ap;
    lean_pin(a.Bh, a.ysrc, a.Vc, a.r, a.Hr, a.z, a.yin0, a.occ_idx, a.occ_fl, a.seed, a.R, a.first, a.pub_fl, a.drop_e, a.n_items, a.dbg);
    LCLK_INIT(a.dbg, 0); LCLK(1);
    LSPAN_BEGIN(a.dbgtile, 1024);
    const float bias = ldu(a.Bh, 4 * (part * D + nc));
    float4 ay;
    if (L0) ay = ld4(a.ysrc + (size_t)min((unsigned)max(item, 0), (unsigned)a.n_items - 1) * IN + 4 * Qy);      // (rows past M hold any staged id: clamped, masked below)
    else ay = ldu4(a.ysrc, 4 * (rowB * IN + 4 * Qy));
    LCLK(2);
    // ---- the state is needed from here
    const LeanState sx = lean_state(mt);
    const int M = sx.M;
    if (a.first && blockIdx.x == 0 && blockIdx.y == 0 && part == 0 && tid == 0) {
        GAS StepState* sw = (GAS StepState*)st_;
        sw->t_b = (long long)(((unsigned long long)(unsigned)mt.w) | ((unsigned long long)(unsigned)cur_in[2 * B + 4] << 32));
        sw->g_b = (long long)(((unsigned long long)(unsigned)mt.x) | ((unsigned long long)(unsigned)mt.y << 32));
        sw->M_b = M;
    }
    const bool rowok = (int)rowA < M, odd = (sx.g & 1u) != 0;
    if (L0) {
        if (!rowok) item = -1;
        if (blockIdx.x == 0 && part == 0 && wid == 0 && lg == 0 && rowA < B) {
            a.occ_idx[rowA] = item;
            if (item >= 0 && a.pub_fl) {      // first / last occurrence of the item in this step's gathered-row list (k_update)
                int* fl = (int*)a.occ_fl + 4 * (size_t)item;
                atomicMax(fl, (int)rowA + 1);
                atomicMax(fl + 1, a.R - (int)rowA);
                atomicAdd(fl + 2, 1);
            }
        }
    }
    if ((int)m0 >= M) return;
    LCLK(3);
    float4 ah = odd ? ah1 : ah0;
    const float hep = odd ? he1 : he0;
    if (DROPE) {
        const float4 mk = drop_mult4(a.seed, sx.g, G4R_STREAM_DROP_EMBED, rowA, Qy, 1.0f - a.drop_e);
        ay.x *= mk.x; ay.y *= mk.y; ay.z *= mk.z; ay.w *= mk.w;
    }
    if (!(oky && rowok)) ay = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(okh && rowok)) ah = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L0 && blockIdx.x == 0 && part == 0 && oky && rowok) stu4(a.yin0, 4 * (rowA * IN + 4 * Qy), ay);
    LCLK(4);
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (part) {      // (uniform; the candidate's input part has no hidden segment)
        acc0 = mfma16(ah.x, bh[0], acc0);
        acc1 = mfma16(ah.y, bh[1], acc1);
        acc0 = mfma16(ah.z, bh[2], acc0);
        acc1 = mfma16(ah.w, bh[3], acc1);
    }
    acc0 = mfma16(ay.x, bx[0], acc0);
    acc1 = mfma16(ay.y, bx[1], acc1);
    acc0 = mfma16(ay.z, bx[2], acc0);
    acc1 = mfma16(ay.w, bx[3], acc1);
    const f32x4 acc = acc0 + acc1;
    LCLK_USE(acc[0]); LCLK(5);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) sJ[(rg * 8 + wid) * 64 + lane] = acc[rg];
    __syncthreads();
    if (wid >= 4) return;
    LCLK(6);
    float v = bias;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += sJ[(wid * 8 + w) * 64 + lane];      // K slices in wave order
    if (ncol >= D || (int)rowe >= M) return;
    const unsigned o = 4 * (rowe * D + ncol);
    if (part == 0) stu(a.Vc, o, v);
    else if (part == 1) { const float rr = sigmoidf_(v); stu(a.r, o, rr); stu(a.Hr, o, hep * rr); }
    else stu(a.z, o, sigmoidf_(v));
    LCLK(7);
    LSPAN_END();
}

// (hipcc emits the device code of a __global__ template only for explicit instantiations)
#define G4R_LEAN_V_ARGS const LeanV*, StepState*, const int*, const float*, const float*, const float*, const float*, unsigned, unsigned
template __global__ void k_gru_v<true, false>(G4R_LEAN_V_ARGS);
template __global__ void k_gru_v<true, true>(G4R_LEAN_V_ARGS);
template __global__ void k_gru_v<false, false>(G4R_LEAN_V_ARGS);

// Forward, launch 2: c = act(Hr Wh + Vc); h = (1 - z) H + z c; hidden dropout; reset switch -> next H; saves c, hd (gru4rec.py:474-479).
// grid (ceil(D / 16), ceil(B / 16)), eight waves: wave w takes super-step w of K = D.  rst_: cur_in + B (reset flags), followed by the state.
__global__ __launch_bounds__(512) void k_gru_h(const LeanH* __restrict__ ap, const float* Wh_, const float* Hr_, const float* Vc_,
                                               const float* z_, const int* rst_, const float* H0_, const float* H1_, unsigned D, unsigned B) {
    __shared__ float sJ[4 * 8 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned m0 = blockIdx.y * 16, rowA = m0 + li;
    const GAS int* rstp = (const GAS int*)rst_;
    const int4 mt = ldi4(rstp + B);
    const unsigned ncol = 16 * blockIdx.x + li, nc = min(ncol, D - 1);
    const unsigned Q = 4 * wid + lg;
    const bool okq = Q < (D >> 2);
    const unsigned Qc = okq ? Q : 0, Db = 4 * D;
    const unsigned offw = 4 * Qc * Db + 4 * nc;
    float4 av = ldu4((const GAS float*)Hr_, 4 * (min(rowA, B - 1) * D + 4 * Qc));
    float bw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bw[u] = ldu((const GAS float*)Wh_, offw + u * Db);
    // epilogue operands of waves 0 .. 3
    const unsigned rowe = m0 + 4 * lg + (wid & 3), roweB = min(rowe, B - 1);
    const unsigned oe = 4 * (roweB * D + nc);
    const float vce = ldu((const GAS float*)Vc_, oe), ze = ldu((const GAS float*)z_, oe);
    const float he0 = ldu((const GAS float*)H0_, oe), he1 = ldu((const GAS float*)H1_, oe);
    const int rst = ldu_i(rstp, 4 * roweB);
    const LeanH a = *ap;
    lean_pin(a.c, a.hd, a.seed, a.hidden_act, a.stream, a.ha_p0, a.ha_p1, a.drop_h, a.dbg);
    LCLK_INIT(a.dbg, 16); LCLK(1);
    LSPAN_BEGIN(a.dbgtile, 1280);
    const LeanState sx = lean_state(mt);
    const int M = sx.M;
    if ((int)m0 >= M) return;
    LCLK(2);
    const bool odd = (sx.g & 1u) != 0;
    const float he = odd ? he1 : he0;
    GAS float* Hnext = (GAS float*)(odd ? H0_ : H1_);
    if (!(okq && (int)rowA < M)) av = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc0 = mfma16(av.x, bw[0], acc0);
    acc1 = mfma16(av.y, bw[1], acc1);
    acc0 = mfma16(av.z, bw[2], acc0);
    acc1 = mfma16(av.w, bw[3], acc1);
    const f32x4 acc = acc0 + acc1;
    LCLK_USE(acc[0]); LCLK(5);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) sJ[(rg * 8 + wid) * 64 + lane] = acc[rg];
    __syncthreads();
    if (wid >= 4) return;
    LCLK(6);
    float v = vce;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += sJ[(wid * 8 + w) * 64 + lane];
    if (ncol >= D || (int)rowe >= M) return;
    const float cc = act_fwd(a.hidden_act, a.ha_p0, a.ha_p1, v);
    float h = (1.0f - ze) * he + ze * cc;
    if (a.drop_h > 0.f) h *= drop_mult(a.seed, sx.g, (unsigned)a.stream, rowe, ncol, 1.0f - a.drop_h);
    const unsigned o = 4 * (rowe * D + ncol);
    stu(a.c, o, cc);
    stu(a.hd, o, h);
    stu(Hnext, o, rst ? 0.f : h);
    LCLK(7);
    LSPAN_END();
}

// ---------------------------------------------------------------------------------------------
// Backward, launch 1 (no BPTT: H is a constant input, gru4rec.py:460-463,576).  Workgroup (K slice j = 16 columns of the layer, 16 rows):
//   dh = split-K slabs of k_score_bwd in fixed order (top layer) / the upper layer's dy, through the hidden-dropout mask;
//   da = dh z act'(c), dz' = dh (c - H) z (1 - z) for its 16 x 16 elements -> dV[:, 0:D], dV[:, 2D:3D];
//   dr'_j = da[:, slice j] Wh[:, slice j]^T for ALL D columns -> partial plane drp[j] (k_gru_dy adds the planes and applies H r (1 - r)).
// Two waves, no LDS, no barrier: each builds the da fragment itself (the same 13 KB of loads) and takes four of the <= 8 column tiles.
#define LN_SLB 18      // slabs per batch of loads (one round trip for the 9 / 17 slabs of k_score_bwd / k_score_b)
__global__ __launch_bounds__(128) void k_gru_da(const LeanDa* __restrict__ ap, const int* meta_, const float* dsrc_, const float* Wh_, const float* z_,
                                                const float* c_, const float* H0_, const float* H1_, unsigned dims, unsigned B) {
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned D = dims & 0xFFFFu;
    const int ks = (int)(dims >> 16);
    const unsigned j = blockIdx.x, m0 = blockIdx.y * 16, row = m0 + li;
    const int4 mt = ldi4((const GAS int*)meta_);
    const unsigned col = 16 * j + 4 * lg;
    const bool colok = col < D;      // (D is a multiple of 4: a quad is inside or outside)
    const unsigned colc = colok ? col : 0;
    const unsigned off = 4 * (min(row, B - 1) * D + colc);
    const unsigned ps = 4 * B * D;
    const GAS float* dsrc = (const GAS float*)dsrc_;
    // the first batch of planes, the gates, the B fragments Wh[n][16 j + 4 lg ..] of the wave's four column tiles: one round trip
    float4 v[LN_SLB];
#pragma unroll
    for (int q = 0; q < LN_SLB; ++q) v[q] = ldu4(dsrc, off + (unsigned)min(q, ks - 1) * ps);
    const float4 z4 = ldu4((const GAS float*)z_, off), c4 = ldu4((const GAS float*)c_, off);
    const float4 h40 = ldu4((const GAS float*)H0_, off), h41 = ldu4((const GAS float*)H1_, off);
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = ldu4((const GAS float*)Wh_, 4 * (min(16 * (4 * wid + q) + li, D - 1) * D + colc));
    const LeanDa a = *ap;
    lean_pin(a.dV, a.drp, a.seed, a.hidden_act, a.stream, a.ha_p0, a.ha_p1, a.drop_h, a.dbg);
    LCLK_INIT((GAS long long*)nullptr, 32); LCLK(1);
    LSPAN_BEGIN(a.dbgtile, 1400);
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < LN_SLB; ++q) {      // planes in fixed order
        const float w = (q < ks) ? 1.f : 0.f;
        dh.x = fmaf(w, v[q].x, dh.x); dh.y = fmaf(w, v[q].y, dh.y); dh.z = fmaf(w, v[q].z, dh.z); dh.w = fmaf(w, v[q].w, dh.w);
    }
    for (int k0 = LN_SLB; k0 < ks; k0 += LN_SLB) {      // more planes than one batch holds
        float4 v2[LN_SLB];
#pragma unroll
        for (int q = 0; q < LN_SLB; ++q) v2[q] = ldu4(dsrc, off + (unsigned)min(k0 + q, ks - 1) * ps);
#pragma unroll
        for (int q = 0; q < LN_SLB; ++q) {
            const float w = (k0 + q < ks) ? 1.f : 0.f;
            dh.x = fmaf(w, v2[q].x, dh.x); dh.y = fmaf(w, v2[q].y, dh.y); dh.z = fmaf(w, v2[q].z, dh.z); dh.w = fmaf(w, v2[q].w, dh.w);
        }
    }
    LCLK_USE(dh.x); LCLK(2);
    const LeanState sx = lean_state(mt);
    const int M = sx.M;
    if ((int)m0 >= M) return;
    LCLK(3);
    const float4 h4 = (sx.g & 1u) ? h41 : h40;
    if (a.drop_h > 0.f) {
        const float4 mk = drop_mult4(a.seed, sx.g, (unsigned)a.stream, row, colc >> 2, 1.0f - a.drop_h);
        dh.x *= mk.x; dh.y *= mk.y; dh.z *= mk.z; dh.w *= mk.w;
    }
    const bool ok = (int)row < M && colok;
    const float hh[4] = {h4.x, h4.y, h4.z, h4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, dd[4] = {dh.x, dh.y, dh.z, dh.w};
    float da[4], dzp[4], ad[4];
    if (a.hidden_act == G4R_ACT_TANH) {      // the default, kept out of the per-element switch
#pragma unroll
        for (int u = 0; u < 4; ++u) ad[u] = 1.0f - cc[u] * cc[u];
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) ad[u] = act_bwd_from_out(a.hidden_act, a.ha_p0, a.ha_p1, cc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float dz = dd[u] * (cc[u] - hh[u]), dc = dd[u] * zz[u];
        da[u] = ok ? dc * ad[u] : 0.f;
        dzp[u] = ok ? dz * zz[u] * (1.f - zz[u]) : 0.f;
    }
    if (wid == 0 && ok) {
        const unsigned ov = 4 * (row * 3 * D + col);
        stu4(a.dV, ov, make_float4(da[0], da[1], da[2], da[3]));
        stu4(a.dV, ov + 8 * D, make_float4(dzp[0], dzp[1], dzp[2], dzp[3]));
    }
    LCLK(4);
    // the product TRANSPOSED (A = the Wh fragment, B = the da fragment: the same registers, operands swapped): a lane then holds four
    // consecutive columns n = 16 tile + 4 lg .. of ONE batch row li -- one 16-byte store per tile instead of four scattered dwords
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = mfma16(bq[q].x, da[0], acc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = mfma16(bq[q].y, da[1], acc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = mfma16(bq[q].z, da[2], acc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = mfma16(bq[q].w, da[3], acc[q]);
    LCLK_USE(acc[0][0]); LCLK(5);
    if ((int)row < M) {
        const unsigned ob = j * ps + 4 * (row * D + 4 * lg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned n = 16 * (4 * wid + q) + 4 * lg;      // (D is a multiple of 4: the four columns are inside or outside)
            if (n < D) stu4(a.drp, ob + 64 * (4 * wid + q), make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]));
        }
    }
    LCLK(6);
    LSPAN_END();
}

// Backward, launch 2: dy tile = [da | dr' | dz'] Wx^T (K = 3 D), 16 rows x 16 input columns per workgroup, sixteen waves over K:
// waves 0 .. 3 the da part, 4 .. 7 the dz' part (two super-steps each, A fragments straight from dV), 8 .. 15 the dr' part (one super-step
// each) -- its A fragment is built here: dr' = (sum of the <= 8 partial planes of k_gru_da, plane order) * H * r (1 - r); column tile 0
// writes it to dV[:, D:2D] for the dense-gradient tiles.  Epilogue (waves 0 .. 3, as k_gru_bwd_b): layer 0 embedding-dropout mask and the
// Adagrad pieces dSx / dAx (or the accumulator in place for a single-occurrence item), else the lower layer's dh.
#define LN_PL 8        // partial planes (= ceil(D / 16) <= 8)
__global__ __launch_bounds__(1024) void k_gru_dy(const LeanDy* __restrict__ ap, const int* meta_, const int* occ_idx_, const float* dV_, const float* drp_,
                                                 const float* Wx_, const float* r_, unsigned dims, unsigned B) {
    __shared__ float sJ[4 * 16 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned D = dims & 0xFFFFu, IN = dims >> 16, Dq = D >> 2;
    const unsigned m0 = blockIdx.y * 16, row = m0 + li, rowB = min(row, B - 1);
    const unsigned ncol = 16 * blockIdx.x + li, nc = min(ncol, IN - 1);
    const unsigned rowe = m0 + 4 * lg + (wid & 3);     // epilogue of waves 0 .. 3: component rg = wave
    const GAS float* Wx = (const GAS float*)Wx_;
#if defined(G4R_CLK_TRACE)
    const long long tk0 = wall_clock64();
    GAS long long* ck = nullptr;
#define DYCK(i) do { if (ck) ck[i] = wall_clock64(); } while (0)
#else
#define DYCK(i)
#endif
    f32x4 acc;
    int4 mt;
    float a2 = 0.f;
    int cnt2 = 0, itm = -1;
    LeanState sx;
    if (wid < 8) {
        // ---- da / dz' part: A fragments straight from dV (k_gru_da wrote them)
        if (wid < 4) itm = ldu_i((const GAS int*)occ_idx_, 4 * min(rowe, B - 1));      // epilogue (layer 0): item of the output row
        mt = ldi4((const GAS int*)meta_);
        const unsigned part = (wid >> 2) * 2;                   // 0 da, 2 dz'
        const unsigned Q0 = 8 * (wid & 3) + lg, Q1 = Q0 + 4;    // two super-steps
        const bool ok0 = Q0 < Dq, ok1 = Q1 < Dq;
        const unsigned Qa = ok0 ? Q0 : 0, Qb = ok1 ? Q1 : 0;
        const unsigned offw = 4 * (nc * 3 * D + part * D), offa = 4 * (rowB * 3 * D + part * D);
        float4 a0 = ldu4((const GAS float*)dV_, offa + 16 * Qa), a1 = ldu4((const GAS float*)dV_, offa + 16 * Qb);
        const float4 b0 = ldu4(Wx, offw + 16 * Qa), b1 = ldu4(Wx, offw + 16 * Qb);
        const LeanDy a = *ap;
        lean_pin(a.accT, a.occ_fl, a.n_items, a.layer0);
#if defined(G4R_CLK_TRACE)
        ck = (a.dbg && blockIdx.x == 1 && blockIdx.y == 1 && tid == 0) ? a.dbg + 32 : nullptr;
        if (ck) ck[0] = tk0;
#endif
        DYCK(1);
        if (a.layer0 && wid < 4) {      // -> the item's accumulator element, its occurrence count
            const unsigned ic = min((unsigned)max(itm, 0), (unsigned)a.n_items - 1);      // (rows past M hold an old id: clamped, unused)
            a2 = a.accT[(size_t)ic * IN + nc];
            cnt2 = a.occ_fl[4 * (size_t)ic + 2];
        }
        DYCK(2);
        sx = lean_state(mt);
        DYCK(3);
        const bool rowok = (int)row < sx.M;
        if (!(ok0 && rowok)) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(ok1 && rowok)) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc0 = mfma16(a0.x, b0.x, acc0);
        acc1 = mfma16(a0.y, b0.y, acc1);
        acc0 = mfma16(a0.z, b0.z, acc0);
        acc1 = mfma16(a0.w, b0.w, acc1);
        acc0 = mfma16(a1.x, b1.x, acc0);
        acc1 = mfma16(a1.y, b1.y, acc1);
        acc0 = mfma16(a1.z, b1.z, acc0);
        acc1 = mfma16(a1.w, b1.w, acc1);
        acc = acc0 + acc1;
#if defined(G4R_CLK_TRACE)
        if (ck) { if (acc[0] == 123.4f) ck[15] = 0; ck[4] = wall_clock64(); }
#endif
    } else {
        // ---- dr' part: planes of k_gru_da -> (sum) * H * r (1 - r)
        mt = ldi4((const GAS int*)meta_);
        const unsigned Q0 = 4 * (wid - 8) + lg;
        const bool ok0 = Q0 < Dq;
        const unsigned Qa = ok0 ? Q0 : 0;
        const unsigned offr = 4 * (rowB * D + 4 * Qa), NTD = (D + 15) >> 4, ps = 4 * B * D;
        float4 pv[LN_PL];
#pragma unroll
        for (int q = 0; q < LN_PL; ++q) pv[q] = ldu4((const GAS float*)drp_, offr + min((unsigned)q, NTD - 1) * ps);
        const float4 r4 = ldu4((const GAS float*)r_, offr);
        const float4 b0 = ldu4(Wx, 4 * (nc * 3 * D + D) + 16 * Qa);
        const LeanDy a = *ap;
        lean_pin(a.H0, a.H1, a.dV);
#if defined(G4R_CLK_TRACE)
        ck = (a.dbg && blockIdx.x == 1 && blockIdx.y == 1 && tid == 512) ? a.dbg + 40 : nullptr;
        if (ck) ck[0] = tk0;
#endif
        DYCK(1);
        const float4 h40 = ldu4(a.H0, offr), h41 = ldu4(a.H1, offr);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < LN_PL; ++q) {      // plane order
            const float w = ((unsigned)q < NTD) ? 1.f : 0.f;
            s.x = fmaf(w, pv[q].x, s.x); s.y = fmaf(w, pv[q].y, s.y); s.z = fmaf(w, pv[q].z, s.z); s.w = fmaf(w, pv[q].w, s.w);
        }
        DYCK(2);
        sx = lean_state(mt);
        const float4 h4 = (sx.g & 1u) ? h41 : h40;
        s.x *= h4.x * r4.x * (1.f - r4.x); s.y *= h4.y * r4.y * (1.f - r4.y);
        s.z *= h4.z * r4.z * (1.f - r4.z); s.w *= h4.w * r4.w * (1.f - r4.w);
        const bool rowok = (int)row < sx.M;
        if (blockIdx.x == 0 && ok0 && rowok) stu4(a.dV, 4 * (row * 3 * D + D + 4 * Qa), s);
        if (!(ok0 && rowok)) s = make_float4(0.f, 0.f, 0.f, 0.f);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc0 = mfma16(s.x, b0.x, acc0);
        acc1 = mfma16(s.y, b0.y, acc1);
        acc0 = mfma16(s.z, b0.z, acc0);
        acc1 = mfma16(s.w, b0.w, acc1);
        acc = acc0 + acc1;
#if defined(G4R_CLK_TRACE)
        if (ck) { if (acc[0] == 123.4f) ck[7] = 0; ck[3] = wall_clock64(); }
#endif
    }
    const int M = sx.M;
    if ((int)m0 >= M) return;      // (uniform over the workgroup: no wave is left at the barrier)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) sJ[(rg * 16 + wid) * 64 + lane] = acc[rg];
    __syncthreads();
    if (wid >= 4) return;
    const LeanDy a = *ap;
    LCLK_INIT(a.dbg, 48); LCLK(6);
    LSPAN_BEGIN(a.dbgtile, 1500);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += sJ[(wid * 16 + w) * 64 + lane];      // K slices in wave order
    if (ncol >= IN || (int)rowe >= M) return;
    const unsigned o = 4 * (rowe * IN + ncol);
    if (a.layer0) {
        if (a.drop_e > 0.f) v *= drop_mult(a.seed, sx.g, G4R_STREAM_DROP_EMBED, rowe, ncol, 1.0f - a.drop_e);
        const float an = a2 + G4R_MUT_ACC(v * v);
        GAS float* dSx = a.dSx + (size_t)(sx.g & (unsigned)a.defer_mask) * (size_t)a.dSx_stride;
        stu(dSx, o, a.generic ? v : G4R_MUT_STEP(a.lr * v * frsq(an + G4R_EPS_ADAGRAD)));
        if (!a.generic && cnt2 == 1 && itm >= 0) a.accT[(size_t)itm * IN + ncol] = an;      // single occurrence: in place (see k_score_bwd)
        else stu(a.dAx, o, an);
    } else {
        stu(a.dylo, o, v);
    }
    LCLK(7);
    LSPAN_END();
}

// ---------------------------------------------------------------------------------------------
// Scoring forward of a narrow top layer (D <= 128) at RSC15-like sizes: Sc[B, N] = h Wy[Y | samples]^T + By - logq lq (gru4rec.py:493-495),
// one 32 x 32 tile per workgroup (272 workgroups at B = 128, N = 2176), four waves = four quarters of K, each holding the WHOLE tile for its
// two super-steps: 4 + 4 float4 fragment loads (the gathered Wy rows of the tile's columns behind their staged item ids: both operands are
// K-contiguous), 32 MFMAs on four accumulators, no operand is loaded twice inside the workgroup (32 KB per workgroup).  Join through LDS,
// wave q finishes sub-tile q (bias - logQ correction of the column's item, row < M, column < N).  Row tile 0 publishes col_item and the
// Y | samples part of occ_idx / occ_fl.  Replaces k_score_fwd's 64 x 32 LDS-staged tiles there (5.0 -> ~2 us in the step, tools/kn_cost.py).
template <bool LOGQ>
__global__ __launch_bounds__(256) void k_score_s(const LeanS* __restrict__ ap, const int* meta_, const int* cur_col_, const float* hd_, const float* Wy_,
                                                 const float* By_, float* Sc_, unsigned dimsA, unsigned dimsB) {
    __shared__ f32x4 sJ[4 * 4 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned D = dimsA & 0xFFFFu, B = dimsA >> 16, N = dimsB & 0xFFFFu, ldSc = dimsB >> 16, Dq = D >> 2;
    const unsigned n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const GAS int* cur_col = (const GAS int*)cur_col_;
    // items of the two column sub-tiles' columns (staged by the previous step's bookkeeping), then the step's (g, M, t)
    const unsigned nA = n0 + li, nB = n0 + 16 + li;
    int it0 = ldu_i(cur_col, 4 * min(nA, ldSc - 1)), it1 = ldu_i(cur_col, 4 * min(nB, ldSc - 1));
    const int4 mt = ldi4((const GAS int*)meta_);
    // A fragments: h rows of the two row sub-tiles, this wave's two super-steps
    const unsigned Q0 = 8 * wid + lg, Q1 = Q0 + 4;
    const bool ok0 = Q0 < Dq, ok1 = Q1 < Dq;
    const unsigned Qa = ok0 ? Q0 : 0, Qb = ok1 ? Q1 : 0;
    const GAS float* hd = (const GAS float*)hd_;
    const unsigned ra = 4 * (min(m0 + li, B - 1) * D), rb = 4 * (min(m0 + 16 + li, B - 1) * D);
    float4 a00 = ldu4(hd, ra + 16 * Qa), a01 = ldu4(hd, ra + 16 * Qb), a10 = ldu4(hd, rb + 16 * Qa), a11 = ldu4(hd, rb + 16 * Qb);
    const LeanS a = *ap;
    lean_pin(a.col_item, a.occ_idx, a.occ_fl, a.R, a.pub_fl, a.logq, a.dbg);
    LCLK_INIT(a.dbg, 56); LCLK(0);
    LSPAN_BEGIN(a.dbgtile, 4096);
    // B fragments: the gathered Wy rows (64-bit row addresses: the table may exceed 4 GB)
    if (nA >= ldSc) it0 = -1;
    if (nB >= ldSc) it1 = -1;
    const GAS float* Wy = (const GAS float*)Wy_;
    const GAS float *w0 = Wy + (size_t)max(it0, 0) * D, *w1 = Wy + (size_t)max(it1, 0) * D;
    float4 b00 = ld4(w0 + 4 * Qa), b01 = ld4(w0 + 4 * Qb), b10 = ld4(w1 + 4 * Qa), b11 = ld4(w1 + 4 * Qb);
    // epilogue operand of wave q (sub-tile (q >> 1, q & 1)): bias - logQ correction of its column's item
    const int ite = (wid & 1) ? it1 : it0;
    const unsigned ne = (wid & 1) ? nB : nA;
    float bias = ((const GAS float*)By_)[max(ite, 0)];
    if (LOGQ) {
        const DevModel& m = *a.mp;
        bias -= a.logq * (ne < B ? m.lq_tgt : m.lq_smp)[max(ite, 0)];
    }
    LCLK(1);
    const LeanState sx = lean_state(mt);
    const int M = sx.M;
    if (blockIdx.y == 0 && wid == 0 && lg < 2) {      // lanes (li, 0): column n0 + li; lanes (li, 1): column n0 + 16 + li
        const unsigned n = lg ? nB : nA;
        const int item = lg ? it1 : it0;
        if (n < ldSc) {
            a.col_item[n] = item;
            if (n < N) {
                a.occ_idx[n] = item;
                if (item >= 0 && a.pub_fl) {
                    int* fl = (int*)a.occ_fl + 4 * (size_t)item;
                    atomicMax(fl, (int)(B + n) + 1);
                    atomicMax(fl + 1, a.R - (int)(B + n));
                    atomicAdd(fl + 2, 1);
                }
            }
        }
    }
    if ((int)m0 >= M) return;
    LCLK(2);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(ok0 && (int)(m0 + li) < M)) a00 = z4;
    if (!(ok1 && (int)(m0 + li) < M)) a01 = z4;
    if (!(ok0 && (int)(m0 + 16 + li) < M)) a10 = z4;
    if (!(ok1 && (int)(m0 + 16 + li) < M)) a11 = z4;
    if (!(ok0 && it0 >= 0)) b00 = z4;
    if (!(ok1 && it0 >= 0)) b01 = z4;
    if (!(ok0 && it1 >= 0)) b10 = z4;
    if (!(ok1 && it1 >= 0)) b11 = z4;
    f32x4 c00 = (f32x4){0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;      // c[row sub-tile][column sub-tile]
#define LS_STEP(A0, A1, B0, B1, C)                                  \
    c00 = mfma16(A0.C, B0.C, c00); c01 = mfma16(A0.C, B1.C, c01);   \
    c10 = mfma16(A1.C, B0.C, c10); c11 = mfma16(A1.C, B1.C, c11);
    LS_STEP(a00, a10, b00, b10, x) LS_STEP(a00, a10, b00, b10, y) LS_STEP(a00, a10, b00, b10, z) LS_STEP(a00, a10, b00, b10, w)
    LS_STEP(a01, a11, b01, b11, x) LS_STEP(a01, a11, b01, b11, y) LS_STEP(a01, a11, b01, b11, z) LS_STEP(a01, a11, b01, b11, w)
#undef LS_STEP
    LCLK_USE(c00[0] + c11[0]); LCLK(3);
    sJ[(0 * 4 + wid) * 64 + lane] = c00; sJ[(1 * 4 + wid) * 64 + lane] = c01;
    sJ[(2 * 4 + wid) * 64 + lane] = c10; sJ[(3 * 4 + wid) * 64 + lane] = c11;
    __syncthreads();
    LCLK(4);
    f32x4 v = sJ[(wid * 4 + 0) * 64 + lane];      // K quarters in wave order
    v += sJ[(wid * 4 + 1) * 64 + lane]; v += sJ[(wid * 4 + 2) * 64 + lane]; v += sJ[(wid * 4 + 3) * 64 + lane];
    if (ne >= N) return;
    const unsigned r0 = m0 + 16 * (wid >> 1) + 4 * lg;
    GAS float* Sc = (GAS float*)Sc_;
    float o[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { o[rg] = v[rg] + bias; lean_keep(o[rg]); }      // (complete before the conditional stores: lean_keep)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
        if ((int)(r0 + rg) < M) stu(Sc, 4 * ((r0 + rg) * ldSc + ne), o[rg]);
    LCLK(5);
    LSPAN_END();
}
template __global__ void k_score_s<false>(const LeanS*, const int*, const int*, const float*, const float*, const float*, float*, unsigned, unsigned);
template __global__ void k_score_s<true>(const LeanS*, const int*, const int*, const float*, const float*, const float*, float*, unsigned, unsigned);

// ---------------------------------------------------------------------------------------------
// Scoring backward of a narrow top layer at RSC15-like sizes (B <= 128, D <= 128), two roles in one launch (block ranges), both on the same
// skeleton: eight waves = eight slices of K, each holding FOUR 16 x 16 accumulators that share one operand fragment -- the row-major operand
// is loaded as float4 along its contiguous dimension, component c feeding sub-tile c (output index 4 i + c: a permutation that leaves every
// lane with 16 CONTIGUOUS outputs, four per sub-tile row) --, one LDS hand-over in the layout [register rg][wave][lane] -> float4 over the four
// sub-tiles, and wave rg finishing one float4 per lane (16-byte loads / stores of accumulator, step and slab rows).
//   role A (blockIdx.x < nA): dSy[n, d] = sum_b ds[b, n] h[b, d] for 16 score columns x 64 d (the column d == D is a ones column of h:
//          dSBy = colsum(ds)); K = the batch, wave w takes rows 16 w .. 16 w + 15.  Epilogue as k_score_bwd role A: per-occurrence Adagrad
//          scaling with the item's PRE-step accumulator (gru4rec.py:335-340), accumulator in place for single-occurrence items.
//   role B: slab kc of dh = ds Sy (gathered Wy rows of 128 score columns), 16 batch rows x 64 d; wave w takes columns 16 w .. 16 w + 15 of
//          the slab.  k_gru_da adds the slabs in slab order.
// Replaces k_score_bwd<32, 128> (LDS-staged 32 x 32 tiles, 6.0 us in the step) where k_score_s replaces k_score_fwd.
__global__ __launch_bounds__(512) void k_score_b(const LeanB* __restrict__ ap, const int* meta_, const int* cur_col_, const float* Sc_, const float* hd_,
                                                 const float* Wy_, float* accWy_, unsigned dimsA, unsigned dimsB) {
    __shared__ f32x4 sJ[4 * 8 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned D = dimsA & 0xFFFFu, B = dimsA >> 16, N = dimsB & 0xFFFFu, ldSc = dimsB >> 16;
    const GAS int* cur_col = (const GAS int*)cur_col_;
    const GAS float *Sc = (const GAS float*)Sc_, *hd = (const GAS float*)hd_;
    const int4 mt = ldi4((const GAS int*)meta_);
    const LeanB a = *ap;      // (its role-dependent fields are used far below: the compiler's lazy loads cost nothing there)
    LSPAN_BEGIN(a.dbgtile, 2048);
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x < (unsigned)a.nA) {
        // ---------------- role A
        const unsigned nt = blockIdx.x / (unsigned)a.ndh, dh_ = blockIdx.x - nt * (unsigned)a.ndh;
        const unsigned n0 = 16 * nt, d0 = 64 * dh_;
        const unsigned n = n0 + li, nc = min(n, ldSc - 1);
        // epilogue operands of wave rg (requested first: the gathers behind the item id are the longest chain of the kernel)
        const unsigned d4 = d0 + 16 * lg + 4 * (wid & 3);
        const int item = ldu_i(cur_col, 4 * nc);
        // operands: wave w -> batch rows 16 w + 4 s + lg, s = 0 .. 3
        float4 av[4];
        float bv[4];
        const unsigned da = d0 + 4 * li;      // h columns da .. da + 3 (A operand: output index i <-> d = d0 + 4 i + c)
        const unsigned dac = min(da, D - 4);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const unsigned b = min(16 * wid + 4 * s_ + lg, B - 1);
            av[s_] = ldu4(hd, 4 * (b * D + dac));
            bv[s_] = ldu(Sc, 4 * (b * ldSc + nc));
        }
        lean_pin(a.accBy, a.occ_fl);
        const bool iok = item >= 0 && n < N;
        const unsigned ic = (unsigned)max(item, 0);
        // (every wave requests them, with clamped addresses and no branch around the loads: a branch here makes hipcc drain ALL loads
        // at its join -- the accumulator gather's round trip then sits in front of the MFMAs instead of under them)
        float4 acc4 = ld4((const GAS float*)accWy_ + (size_t)ic * D + min(d4, D - 4));
        const float accb = a.accBy[ic];
        const int cnt = a.occ_fl[4 * (size_t)ic + 2];
        if (d4 >= D) acc4 = make_float4(accb, 0.f, 0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);      // (all loads out before the first MFMA)
        const LeanState sx = lean_state(mt);
        const int M = sx.M;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const unsigned b = 16 * wid + 4 * s_ + lg;
            const bool bok = (int)b < M;
            float4 h4 = av[s_];
            // columns past the layer: the ones column at d == D (bias gradient), zeros behind it
            const float hx[4] = {h4.x, h4.y, h4.z, h4.w};
            float hv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) hv[c] = (da + c < D) ? hx[c] : ((da + c == D) ? 1.f : 0.f);
            const float dsv = (bok && n < ldSc) ? bv[s_] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma16(hv[c], dsv, acc[c]);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sJ[(rg * 8 + wid) * 64 + lane] = (f32x4){acc[0][rg], acc[1][rg], acc[2][rg], acc[3][rg]};
        __syncthreads();
        if (wid >= 4) return;
        f32x4 g4 = sJ[(wid * 8 + 0) * 64 + lane];      // batch slices in wave order
#pragma unroll
        for (int w = 1; w < 8; ++w) g4 += sJ[(wid * 8 + w) * 64 + lane];
        if (n >= N || d4 > D) return;
        const float gg[4] = {g4[0], g4[1], g4[2], g4[3]}, a0[4] = {acc4.x, acc4.y, acc4.z, acc4.w};
        float st[4], an[4];
        const bool generic = a.generic != 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            an[c] = a0[c] + G4R_MUT_ACC(gg[c] * gg[c]);
            st[c] = iok ? G4R_MUT_ROW(n, G4R_MUT_STEP(a.lr * gg[c] * frsq(an[c] + G4R_EPS_ADAGRAD))) : 0.f;
            if (generic) st[c] = iok ? gg[c] : 0.f;      // raw per-occurrence gradient: the update kernel applies the rule
        }
        const bool single = !generic && iok && cnt == 1;
        const size_t slot = (size_t)(sx.g & (unsigned)a.defer_mask);
        if (d4 < D) {
            st4(a.dSy + slot * (size_t)a.dSy_stride + (size_t)n * D + d4, make_float4(st[0], st[1], st[2], st[3]));
            if (single) st4((GAS float*)accWy_ + (size_t)ic * D + d4, make_float4(an[0], an[1], an[2], an[3]));
            else st4(a.dAy + (size_t)n * D + d4, make_float4(an[0], an[1], an[2], an[3]));
        } else {      // d4 == D: the bias column
            (a.dSBy + slot * (size_t)a.dSBy_stride)[n] = st[0];
            if (single) a.accBy[ic] = an[0]; else a.dABy[n] = an[0];
        }
        LSPAN_END();
        return;
    }
    // ---------------- role B
    {
        const unsigned w_ = blockIdx.x - (unsigned)a.nA;
        const unsigned per = (unsigned)(a.nrb * a.ndb);
        const unsigned kc = w_ / per, rem = w_ - kc * per, rb = rem / (unsigned)a.ndb, db = rem - rb * (unsigned)a.ndb;
        const unsigned b0 = 16 * rb, d0 = 64 * db;
        // wave w: score columns kbeg + 16 w + 4 lg + u (u = 0 .. 3) as K; A operand: ds[b0 + li][those four] (K-contiguous float4),
        // B operand: the gathered Wy rows of the four columns, float4 along d (output index j <-> d = d0 + 4 j + c)
        const unsigned nk = 128 * kc + 16 * wid + 4 * lg;
        const unsigned nkc = min(nk, ldSc - 4);
        const int4 it4 = ldi4(cur_col + nkc);
        float4 ds4 = ldu4(Sc, 4 * (min(b0 + li, B - 1) * ldSc + nkc));
        const unsigned dq = d0 + 4 * li, dqc = min(dq, D - 4);
        const GAS float* Wy = (const GAS float*)Wy_;
        const int its[4] = {it4.x, it4.y, it4.z, it4.w};
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = ld4(Wy + (size_t)max(its[u], 0) * D + dqc);
        __builtin_amdgcn_sched_barrier(0);      // (all four gathers out before the first MFMA: hipcc otherwise sinks two of them behind it)
        const LeanState sx = lean_state(mt);
        const int M = sx.M;
        if ((int)b0 >= M) return;
        const float dsx[4] = {ds4.x, ds4.y, ds4.z, ds4.w};
        const bool bok = (int)(b0 + li) < M;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool kok = nk + u < ldSc && nk == nkc && its[u] >= 0 && dq < D;
            const float av_ = bok ? dsx[u] : 0.f;
            const float4 w4 = kok ? wv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[0] = mfma16(av_, w4.x, acc[0]);
            acc[1] = mfma16(av_, w4.y, acc[1]);
            acc[2] = mfma16(av_, w4.z, acc[2]);
            acc[3] = mfma16(av_, w4.w, acc[3]);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sJ[(rg * 8 + wid) * 64 + lane] = (f32x4){acc[0][rg], acc[1][rg], acc[2][rg], acc[3][rg]};
        __syncthreads();
        if (wid >= 4) return;
        f32x4 v = sJ[(wid * 8 + 0) * 64 + lane];      // column groups in wave order
#pragma unroll
        for (int w = 1; w < 8; ++w) v += sJ[(wid * 8 + w) * 64 + lane];
        const unsigned b = b0 + 4 * lg + (wid & 3);
        if ((int)b < M && dq < D) st4(a.dhpart + ((size_t)kc * B + b) * D + dq, make_float4(v[0], v[1], v[2], v[3]));
        LSPAN_END();
    }
}

// ---------------------------------------------------------------------------------------------
// The step's last launch for narrow layers on one GPU (B <= 128, rows of <= 256 floats, Adagrad(+momentum), no deferral): k_update's three
// roles -- bookkeeping workgroup, dense-gradient tiles with the fused dense Adagrad, per-occurrence sparse row update -- rewritten on the
// rules of this file.  Replaces k_update<1, 32, MOM> there (7.5 us; the sparse role's span was set by owners of repeated items: 3-4 us in
// the LDS-staged list scan behind a workgroup barrier, tools/clk.py).
//   dense tile (16 rows x 64 columns of dWx / dWh / dWrz / dBh, K = the batch over eight waves): the transposed product of k_score_b's role
//     A -- dV as float4 along its columns (A operand), X as dwords (B operand) -- leaves every lane 16 contiguous columns of one output
//     row: accumulator / parameter (/ velocity) quads in, Adagrad, quads out.  gru4rec.py:330-334,390-406.
//   sparse role (one wave per occurrence k of X | Y | samples, eight per workgroup, NO LDS, no barrier): item id and step row with the
//     first loads; behind the id the item's (last, first, count) entry, parameter row and bias -- one round trip; the wave of the LAST
//     occurrence owns the row.  Single occurrences (~90 %) finish right there.  An owner of a repeated item finds the earlier
//     occurrences itself: the id list is 9 KB and L2 resident, so it reads the slice [first, k) with 16-byte loads (up to 1024 ids per
//     round trip), ballots the matches and adds their step rows in occurrence order, eight rows per round trip -- the arithmetic and the
//     order of sparse_update_block (gru4rec.py:335-340,407-431: increments accumulate, accumulator and velocity take the last
//     occurrence's value).  Items whose occurrences are all sampled negatives take the (count - 1) x own row shortcut as there.
template <bool MOM>
__device__ __forceinline__ void lean_rows_update(const LeanU& a, const GAS int* occ_idx, GAS int* occ_fl, const GAS float* dSx, const GAS float* dSy,
                                                 const GAS float* dSBy, unsigned k, unsigned R, unsigned B) {
    const unsigned lane = threadIdx.x & 63;
    const bool tableE = k < B && !a.constrained;
    const unsigned W = tableE ? (unsigned)a.wE : (unsigned)a.wY, nc4 = W >> 2;
    const unsigned c4 = min(lane, nc4 - 1);
    const bool lok = lane < nc4;
    const unsigned kc = min(k, R - 1);
    // first loads: the occurrence's item, its step row, its bias step
    int item = ldu_i(occ_idx, 4 * kc);
    const GAS float* srow = (kc < B) ? dSx + (size_t)kc * W : dSy + (size_t)(kc - B) * W;
    const float4 sk = ld4(srow + 4 * c4);
    const bool bias = k >= B;
    const float bsk = bias ? dSBy[kc - B] : 0.f;
    lean_pin(a.Wy, a.E, a.accWy, a.accE, a.By, a.accBy, a.dAx, a.dAy, a.dABy, a.n_items, a.lr, a.mom, a.lmbd);
    if (k >= R) item = -1;
    GAS float* P = tableE ? a.E : a.Wy;
    GAS float* A = tableE ? a.accE : a.accWy;
    GAS float* V = tableE ? a.velE : a.velWy;
    const unsigned ic = (unsigned)max(item, 0);
    // second level: the item's entry, its parameter (velocity) row, its bias state
    GAS int* flp = occ_fl + 4 * ((tableE ? (size_t)a.n_items : 0) + ic);
    const int4 fl = ldi4(flp);
    const float4 pz = ld4(P + (size_t)ic * W + 4 * c4);
    float4 vz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MOM) vz = ld4(V + (size_t)ic * W + 4 * c4);
    const float bpz = a.By[ic];
    float bvz = 0.f;
    if (MOM) bvz = a.velBy[ic];
    const bool owner = item >= 0 && fl.x == (int)k + 1;
    if (!owner) return;      // wave-uniform
    if (lane == 0) *(GAS int4*)flp = make_int4(0, 0, 0, 0);      // the entry is taken back for the next step
    const float lr = a.lr, momc = a.mom, lmbd = a.lmbd;
    const int n = fl.z;
    const int lo = (a.constrained || k < B) ? 0 : (int)B;
    const int first_j = max(lo, (int)R - fl.y);
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    float Sb = 0.f;
    int nb_e = 0;
    if (n > 1) {
        if (first_j >= 2 * (int)B) {
            // all occurrences are sampled negatives of this step: their score columns are copies of one another, so are their step rows
            for (int cdup = 1; cdup < n; ++cdup) { S.x += sk.x; S.y += sk.y; S.z += sk.z; S.w += sk.w; Sb += bsk; }
            nb_e = n - 1;
        } else {
            // earlier occurrences in [first_j, k): ids in slices of 1024 (four 16-byte loads per lane), matches in ascending order
            for (int base0 = first_j & ~3; base0 < (int)k; base0 += 1024) {
                int4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = ldi4(occ_idx + min(base0 + 256 * u + 4 * (int)lane, (int)((R + 3) & ~3u) - 4));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j0 = base0 + 256 * u + 4 * (int)lane;
                    const int ids[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
                    unsigned long long mk[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) mk[e] = __ballot(ids[e] == item && j0 + e >= first_j && j0 + e < (int)k);
                    unsigned long long any = mk[0] | mk[1] | mk[2] | mk[3];
                    // up to eight matches per batch of row loads (wave-uniform positions in scalar registers)
                    while (any) {
                        int js[8], nj = 0;
                        while (any && nj < 8) {
                            const int l = __builtin_ctzll(any);
                            bool more = false;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if ((mk[e] >> l) & 1ull) {
                                    if (nj < 8) {
                                        const int jv = base0 + 256 * u + 4 * l + e;
#pragma unroll
                                        for (int q = 0; q < 8; ++q) if (q >= nj) js[q] = jv;      // (slots past nj repeat a valid position; no indexed register access)
                                        ++nj; mk[e] &= ~(1ull << l);
                                    } else more = true;
                                }
                            }
                            if (!more) any &= ~(1ull << l);
                        }
                        float4 g[8];
                        float gb[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int jj = js[q];
                            const GAS float* r2 = (jj < (int)B) ? dSx + (size_t)jj * W : dSy + (size_t)(jj - (int)B) * W;
                            g[q] = ld4(r2 + 4 * c4);
                            gb[q] = (bias && jj >= (int)B) ? dSBy[jj - (int)B] : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            if (q < nj) {      // wave-uniform
                                S.x += g[q].x; S.y += g[q].y; S.z += g[q].z; S.w += g[q].w;
                                if (js[q] >= (int)B) { Sb += gb[q]; ++nb_e; }
                            }
                        }
                    }
                }
            }
        }
    }
    // final row: P = P0 - (S + s_k + n reg)   (momentum: V = mom V0 - (s_k + reg), P = P0 + n mom V0 - (S + s_k + n reg))
    const float fn = (float)n;
    const float p0[4] = {pz.x, pz.y, pz.z, pz.w}, v0[4] = {vz.x, vz.y, vz.z, vz.w}, sl[4] = {sk.x, sk.y, sk.z, sk.w};
    const float ss[4] = {S.x + sk.x, S.y + sk.y, S.z + sk.z, S.w + sk.w};
    float pn[4], vn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float reg = (lmbd > 0.f) ? lr * lmbd * p0[e] : 0.f;
        const float tot = (lmbd > 0.f) ? ss[e] + fn * reg : ss[e];
        if (MOM) { vn[e] = momc * v0[e] - (sl[e] + reg); pn[e] = p0[e] + (fn * (momc * v0[e]) - tot); }
        else { vn[e] = 0.f; pn[e] = p0[e] - tot; }
    }
    float bn = 0.f, bvn = 0.f;
    if (bias) {
        const float fb = (float)(nb_e + 1);
        const float reg = (lmbd > 0.f) ? lr * lmbd * bpz : 0.f;
        const float sb = Sb + bsk;
        const float tot = (lmbd > 0.f) ? sb + fb * reg : sb;
        if (MOM) { bn = bpz + (fb * (momc * bvz) - tot); bvn = momc * bvz - (bsk + reg); }
        else bn = bpz - tot;
    }
    // the last occurrence's accumulator row (repeated items only: a single's accumulator was written in place by the producer of its step row)
    float4 ak = make_float4(0.f, 0.f, 0.f, 0.f);
    float bak = 0.f;
    if (n > 1) {
        const GAS float* arow = (k < B) ? a.dAx + (size_t)k * W : a.dAy + (size_t)(k - B) * W;
        ak = ld4(arow + 4 * c4);
        if (bias) bak = a.dABy[k - B];
    }
    lean_keep(bn); lean_keep(bvn); lean_keep(bak);
    const size_t o = (size_t)item * W + 4 * c4;
    if (lok) {
        st4(P + o, make_float4(pn[0], pn[1], pn[2], pn[3]));
        if (MOM) st4(V + o, make_float4(vn[0], vn[1], vn[2], vn[3]));
        if (n > 1) st4(A + o, ak);
    }
    if (bias && lane == 0) {
        a.By[item] = bn;
        if (MOM) a.velBy[item] = bvn;
        if (n > 1) a.accBy[item] = bak;
    }
}

template <bool MOM>
__device__ __forceinline__ void lean_dense_tile(const LeanU& a, const DenseTile* tiles_, unsigned tile, unsigned B) {
    __shared__ f32x4 sJ[4 * 8 * 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GAS DenseTile* tiles = (const GAS DenseTile*)tiles_;
    const DenseTile tl = tiles[tile];      // 16 rows (r0 ..) x 64 columns (c0 ..), fully resolved on the host
    // (g, M) of THIS step out of StepState::*_b: the staged copy behind cur_in is being rewritten for the next step by this launch's own
    // bookkeeping workgroup
    const GAS StepState* sg = (const GAS StepState*)a.st;
    const long long g_ = sg->g_b;
    const int M = sg->M_b;
    const unsigned row = tl.r0 + li, rowc = min(row, (unsigned)tl.nrows - 1);
    const unsigned cq = tl.c0 + 4 * li, cqc = min(cq, (unsigned)tl.ncols - 4);      // A operand: dV columns cq .. cq + 3 (output index i <-> column c0 + 4 i + c)
    // both parities of X (H ping-pong) are requested; gather = 1: the step's input rows as the GRU saw them (yin0); X0 == null: the ones row of dBh
    const bool ones = tl.X0 == nullptr && !tl.gather;
    const GAS float* Xa = tl.gather ? a.yin0 : (ones ? tl.dV : tl.X0);
    const GAS float* Xb = tl.gather ? a.yin0 : (ones ? tl.dV : tl.X1);
    float4 av[4];
    float x0[4], x1[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const unsigned b = min(16 * wid + 4 * s_ + lg, B - 1);
        av[s_] = ldu4(tl.dV, 4 * (b * (unsigned)tl.ldv + (unsigned)tl.coff + cqc));
        const unsigned ox = ones ? 0u : 4 * (b * (unsigned)tl.ldx + rowc);
        x0[s_] = ldu(Xa, ox); x1[s_] = ldu(Xb, ox);
    }
    // epilogue operands of wave rg: accumulator / parameter (/ velocity) quads of (row r0 + li, columns c0 + 16 lg + 4 rg ..)
    const unsigned ce = tl.c0 + 16 * lg + 4 * (wid & 3), cec = min(ce, (unsigned)tl.ncols - 4);
    const size_t off = (size_t)tl.base + (size_t)rowc * tl.ldo + cec;
    const float4 acc4 = ld4(a.dense_acc + off), p4 = ld4(a.dense_p + off);
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MOM) v4 = ld4(a.dense_vel + off);
    __builtin_amdgcn_sched_barrier(0);
    const bool odd = (g_ & 1) != 0;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const unsigned b = 16 * wid + 4 * s_ + lg;
        const bool bok = (int)b < M;
        float xv = ones ? 1.f : (odd ? x1[s_] : x0[s_]);
        if (!(bok && row < (unsigned)tl.nrows)) xv = 0.f;
        const float4 d4 = (cq < (unsigned)tl.ncols) ? av[s_] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[0] = mfma16(d4.x, xv, acc[0]);
        acc[1] = mfma16(d4.y, xv, acc[1]);
        acc[2] = mfma16(d4.z, xv, acc[2]);
        acc[3] = mfma16(d4.w, xv, acc[3]);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) sJ[(rg * 8 + wid) * 64 + lane] = (f32x4){acc[0][rg], acc[1][rg], acc[2][rg], acc[3][rg]};
    __syncthreads();
    if (wid >= 4) return;
    f32x4 g4 = sJ[(wid * 8 + 0) * 64 + lane];      // batch slices in wave order
#pragma unroll
    for (int w = 1; w < 8; ++w) g4 += sJ[(wid * 8 + w) * 64 + lane];
    if (row >= (unsigned)tl.nrows || ce >= (unsigned)tl.ncols) return;
    const float gg[4] = {g4[0], g4[1], g4[2], g4[3]}, a0[4] = {acc4.x, acc4.y, acc4.z, acc4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    float an[4], pn[4], vn[4];
    const float lr = a.lr, momc = a.mom, lmbd = a.lmbd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {      // gru4rec.py:330-334,390-406
        an[c] = a0[c] + G4R_MUT_DACC(gg[c] * gg[c]);
        const float gs = gg[c] * frsq(an[c] + G4R_EPS_ADAGRAD);
        if (MOM) { vn[c] = momc * vv[c] - lr * (gs + lmbd * pp[c]); pn[c] = pp[c] + vn[c]; }
        else { vn[c] = 0.f; pn[c] = pp[c] * (1.0f - lr * lmbd) - lr * gs; }
    }
    st4(a.dense_acc + off, make_float4(an[0], an[1], an[2], an[3]));
    st4(a.dense_p + off, make_float4(pn[0], pn[1], pn[2], pn[3]));
    if (MOM) st4(a.dense_vel + off, make_float4(vn[0], vn[1], vn[2], vn[3]));
}

// Step bookkeeping over 1 + ceil(ldSc / 512) workgroups (k_update's single bookkeeping workgroup took 4.1 us -- the longest path of the
// launch, tools/clk_lean.py: state -> M of the next step -> row of in_idx -> columns in passes, with a lazily loaded descriptor field in
// front of each): part 0 folds the row losses into loss_steps[t] (cost = sum_i L_i / batch_size, gru4rec.py:577; NaN flag, :626),
// advances StepState::*_a and stages the next step's in_idx row, reset flags and (g, M, t); part p >= 1 stages 512 entries of the next
// step's column -> item list (targets | -1 | its row of the sample store | -1: stage_step_inputs' rule).  Every part reads the state
// and then has ONE round trip of loads.
__device__ __forceinline__ void lean_bookkeep(const LeanU& a, unsigned part, unsigned B) {
    const DevModel& m = *a.mp;
    const unsigned tid = threadIdx.x;
    const GAS StepState* sg = (const GAS StepState*)a.st;
    const long long t = sg->t_b, g = sg->g_b;
    const int M = sg->M_b;
    const GAS int *in_idx = m.in_idx, *out_idx = m.out_idx, *Mplan = m.Mplan, *ST = m.ST;
    const GAS unsigned char* reset = m.reset;
    const int gl = m.gl, ns = m.ns, N = m.N, ld = m.ldSc;
    GAS int *ci = m.cur_in, *cc = m.cur_col;
    GAS float* loss_steps = m.loss_steps;
    const GAS float* lossrow = m.lossrow;
    const float inv_B = m.inv_B;
    lean_pin(in_idx, out_idx, Mplan, ST, reset, gl, ns, N, ld, ci, cc, loss_steps, lossrow, inv_B);
    const long long t1 = t + 1, g1 = g + 1;
    const int Mn = Mplan[t1];      // (the plan carries one trailing entry and one trailing row)
    if (part == 0) {
        if (tid < 64) {
            float s_ = 0.f;
            for (int i = (int)tid; i < M; i += 64) s_ += lossrow[i];
            s_ = wave_sum(s_);
            if (tid == 0) {
                const float cost = s_ * inv_B;
                loss_steps[t] = cost;
                GAS StepState* sw = (GAS StepState*)a.st;
                if (isnan(cost)) sw->nan_flag = 1;
                sw->t_a = t1; sw->g_a = g1; sw->M_a = Mn;
                ci[2 * B] = (int)(unsigned)g1; ci[2 * B + 1] = (int)(g1 >> 32); ci[2 * B + 2] = Mn; ci[2 * B + 3] = (int)(unsigned)t1; ci[2 * B + 4] = (int)(t1 >> 32);
            }
        } else if (tid - 64 < B) {
            const unsigned b = tid - 64;
            ci[b] = in_idx[t1 * B + b];
            ci[B + b] = reset[t1 * B + b];
        }
        return;
    }
    const int n = 512 * (int)(part - 1) + (int)tid;
    const int vo = out_idx[t1 * B + min(n, (int)B - 1)];
    const int vs = (ns > 0) ? ST[(size_t)(gl > 0 ? g1 % gl : 0) * ns + min(max(n - (int)B, 0), ns - 1)] : -1;
    if (n < ld) cc[n] = (n < Mn) ? vo : (n >= (int)B && n < N && Mn > 0) ? vs : -1;      // Mn = 0: padding step of a multi-rank plan, nothing is touched
}

// workgroups [0, nbk): bookkeeping; [nbk, nbk + ntiles): dense tiles; the rest: eight occurrences each.
// packA = ntiles | nblk << 16, packB = R | B << 16, nbk = 1 + ceil(ldSc / 512).
template <bool MOM>
__global__ __launch_bounds__(512) void k_update_l(const LeanU* __restrict__ ap, const DenseTile* __restrict__ tiles_, const int* occ_idx_, int* occ_fl_,
                                                  const float* dSx_, const float* dSy_, const float* dSBy_, unsigned packA, unsigned packB, unsigned nbk) {
    const unsigned ntiles = packA & 0xFFFFu, nblk = packA >> 16, R = packB & 0xFFFFu, B = packB >> 16;
    const LeanU a = *ap;
    LSPAN_BEGIN(a.dbgtile, 2700);
    if (blockIdx.x < nbk) { lean_bookkeep(a, blockIdx.x, B); LSPAN_END(); return; }
    const unsigned b = blockIdx.x - nbk;
    if (b < ntiles) { lean_dense_tile<MOM>(a, tiles_, b, B); LSPAN_END(); return; }
    // occurrences strided over the workgroups (wave w of workgroup q takes k = w nblk + q: the owners of the popular items -- the LAST
    // occurrences, with their duplicate sums -- sit together at the end of the list; contiguous, they would share a few workgroups)
    const unsigned q = b - ntiles, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lean_rows_update<MOM>(a, (const GAS int*)occ_idx_, (GAS int*)occ_fl_, (const GAS float*)dSx_, (const GAS float*)dSy_, (const GAS float*)dSBy_, wid * nblk + q, R, B);
    LSPAN_END();
}
template __global__ void k_update_l<false>(const LeanU*, const DenseTile*, const int*, int*, const float*, const float*, const float*, unsigned, unsigned, unsigned);
template __global__ void k_update_l<true>(const LeanU*, const DenseTile*, const int*, int*, const float*, const float*, const float*, unsigned, unsigned, unsigned);
