// Generic LDS-staged fp32 MFMA tile GEMM for the step kernels (gfx950).
//
// Every GEMM of the GRU4Rec step is small (<= a few hundred MFLOP) and its operands live in L2 / Infinity
// Cache; what bounds it is (a) how many CUs pull operands concurrently (one CU sustains only ~25-50 GB/s of
// fine-grained loads) and (b) how many loads each wave has in flight.  So: small output tiles (BM x BN) so that
// every GEMM spreads over >= 100 workgroups, operand chunks copied global -> LDS with 16-byte coalesced loads
// that are ALL issued before the first one is consumed, MFMA fragments read back conflict-free:
//
//   A in memory as [m][k] ("MK") -> sA[m][k], row stride BK+2  (ld/2 odd  => ds_read_b32 fragments conflict-free)
//   A in memory as [k][m] ("KM") -> sA[k][m], row stride BM+16 (ld == 16 mod 32 => conflict-free)
//   B in memory as [k][n] ("KN") -> sB[k][n], row stride BN+16 ; B as [n][k] ("NK") -> sB[n][k], row stride BK+2
//
// Operand providers are functors  float4 load(kk, row, col)  that return 4 consecutive elements of the staging
// tile (row, col..col+3) for the K-chunk starting at kk (zero outside the matrix), so gathers (embedding rows),
// concatenations (y | H) and dropout are fused into the copy.  The epilogue functor epi(m, n, value) receives
// every output element (16 consecutive n per 16 lanes => 64-byte store segments).
#pragma once
#include "g4r_device.cuh"

#ifndef GT_KU
#define GT_KU 8      // k-steps (of 4) per unrolled group of the MFMA loop
#endif

template <int BM, int BN, int BK, bool AKM, bool BNK>
struct TileCfg {
    static constexpr int A_ROWS = AKM ? BK : BM, A_COLS = AKM ? BM : BK;
    static constexpr int LDA = AKM ? (BM + 16) : (BK + 2);
    static constexpr int B_ROWS = BNK ? BN : BK, B_COLS = BNK ? BK : BN;
    static constexpr int LDB = BNK ? (BK + 2) : (BN + 16);
    static constexpr int SMEM_FLOATS = A_ROWS * LDA + B_ROWS * LDB;
    static constexpr int NSUB = (BM / 16) * (BN / 16) / 4;    // 16x16 sub-tiles per wave (4 waves)
    static_assert(BM % 32 == 0 && BN % 32 == 0 && BK % 32 == 0, "tile sizes");
    static_assert(((BM / 16) * (BN / 16)) % 4 == 0, "sub-tiles must divide over 4 waves");
};

// Staging is split in two halves so that the global loads of a chunk are issued long before they are needed:
// stage_issue puts all loads of the chunk in flight (registers), stage_commit writes them to LDS.
template <int ROWS, int COLS, int NTH = 256>
struct StageRegs { static constexpr int NQ = (ROWS * (COLS / 4) + NTH - 1) / NTH; float4 v[NQ]; };

template <int ROWS, int COLS, int NTH, class Load>
__device__ __forceinline__ void stage_issue(StageRegs<ROWS, COLS, NTH>& r, int kk, Load load, int tid) {
    constexpr int C4 = COLS / 4, TOTAL = ROWS * C4;
    if constexpr (NTH % C4 == 0) {
        // a thread keeps its column for all of its loads and only steps through rows: everything the provider derives
        // from the column (which source, bounds, dropout column) is computed once
        constexpr int RSTEP = NTH / C4;
        const int c = 4 * (tid % C4), r0 = tid / C4;
#pragma unroll
        for (int q = 0; q < StageRegs<ROWS, COLS, NTH>::NQ; ++q)
            r.v[q] = load(kk, min(r0 + q * RSTEP, ROWS - 1), c);      // rows past the tile re-load the last row, not stored
    } else {
#pragma unroll
        for (int q = 0; q < StageRegs<ROWS, COLS, NTH>::NQ; ++q) {
            // no branch around the load (a conditional definition of v[q] would force an s_waitcnt vmcnt(0) per load):
            // threads beyond the tile re-load its last element and simply do not store it
            const int e = min(tid + NTH * q, TOTAL - 1);
            r.v[q] = load(kk, e / C4, 4 * (e % C4));
        }
    }
}
// `fix(kk, row, col, v)` (optional) post-processes a staged quad on its way to LDS: providers whose masking / dropout would
// otherwise sit between the loads (and make the compiler wait for every load separately) return the raw quad and do that
// work here, after all loads of the chunk have been issued.
struct NoFix { __device__ __forceinline__ float4 operator()(int, int, int, float4 v) const { return v; } };
template <int ROWS, int COLS, int LD, int NTH, class Fix = NoFix>
__device__ __forceinline__ void stage_commit(float* s, const StageRegs<ROWS, COLS, NTH>& r, int tid, int kk = 0, Fix fix = Fix()) {
    constexpr int C4 = COLS / 4, TOTAL = ROWS * C4;
#pragma unroll
    for (int q = 0; q < StageRegs<ROWS, COLS, NTH>::NQ; ++q) {
        int row, col;
        bool ok;
        if constexpr (NTH % C4 == 0) { row = tid / C4 + q * (NTH / C4); col = 4 * (tid % C4); ok = row < ROWS; }
        else { const int e = tid + NTH * q; row = e / C4; col = 4 * (e % C4); ok = e < TOTAL; }
        if (ok) {
            float* d = s + row * LD + col;
            const float4 v = fix(kk, row, col, r.v[q]);
            if constexpr (LD % 4 == 0) {
                *reinterpret_cast<float4*>(d) = v;
            } else {
                reinterpret_cast<float2*>(d)[0] = make_float2(v.x, v.y);
                reinterpret_cast<float2*>(d)[1] = make_float2(v.z, v.w);
            }
        }
    }
}

// One workgroup (256 threads = 4 waves) computes the BM x BN tile at (m0, n0) over K.  smem: TileCfg::SMEM_FLOATS.
// pre(m, n) -> float4 fetches whatever the epilogue needs besides the accumulator (bias, gates, optimizer state)
// BEFORE the K loop, so those loads fly together with the operand staging instead of costing a round trip at the end.
struct NoPre { __device__ __forceinline__ float4 operator()(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); } };
// optional hook called by all threads once a K chunk [kk, kk + kend) of the operand tiles sits in LDS (sA, row stride LDA)
struct NoHook { __device__ __forceinline__ void operator()(const float*, int, int) const {} };

// NTH = 256: 4 waves, one (BM/16 x BN/16)/4 share of the 16x16 sub-tiles each.  NTH = 512: two such wave groups that
// split every K chunk between them (so 2 waves per SIMD overlap their load-issue and MFMA latencies: the GEMMs of the
// step only occupy a few dozen CUs, one workgroup each) and add their accumulators through LDS at the end; the
// epilogue and its `pre` loads run on group 0.
template <int BM, int BN, int BK, bool AKM, bool BNK, int NTH = 256, class ALoad, class BLoad, class Pre, class Epi, class Hook = NoHook,
          class AFix = NoFix, class BFix = NoFix>
__device__ __forceinline__ void gemm_tile(int m0, int n0, int K, ALoad aload, BLoad bload, Pre pre, Epi epi, float* smem,
                                          GAS long long* clk = nullptr, Hook hook = Hook(), AFix afix = AFix(), BFix bfix = BFix()) {      // clk: optional phase timestamps (debug)
    using C = TileCfg<BM, BN, BK, AKM, BNK>;
    static_assert(NTH == 256 || NTH == 512, "4 or 8 waves");
    float* sA = smem;
    float* sB = smem + C::A_ROWS * C::LDA;
    const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, li = lane & 15, lg = lane >> 4;
    const int grp = (NTH == 512) ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;      // wave-uniform: keeps the K loop scalar
    constexpr int NT = BN / 16;
    f32x4 acc[C::NSUB];
    float4 pf[C::NSUB][4];
    StageRegs<C::A_ROWS, C::A_COLS, NTH> ra;
    StageRegs<C::B_ROWS, C::B_COLS, NTH> rb;
    // chunk 0 operands first, then the epilogue operands: the (older) operand loads can be waited for with a
    // counted vmcnt while the epilogue loads are still in flight
    stage_issue(ra, 0, aload, tid);
    stage_issue(rb, 0, bload, tid);
#pragma unroll
    for (int q = 0; q < C::NSUB; ++q) {
        acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) pf[q][rg] = pre(m0 + ms * 16 + 4 * lg + rg, n0 + ns * 16 + li);   // (both groups: branch-free)
    }
    if (clk && tid == 0) clk[0] = wall_clock64();      // loads issued
    stage_commit<C::A_ROWS, C::A_COLS, C::LDA>(sA, ra, tid, 0, afix);
    stage_commit<C::B_ROWS, C::B_COLS, C::LDB>(sB, rb, tid, 0, bfix);
    __syncthreads();
    if (clk && tid == 0) clk[1] = wall_clock64();      // first chunk in LDS
    for (int kk = 0; kk < K; kk += BK) {
        const bool more = kk + BK < K;
        if (more) {         // next chunk's loads fly during this chunk's MFMAs
            stage_issue(ra, kk + BK, aload, tid);
            stage_issue(rb, kk + BK, bload, tid);
        }
        const int kend = min(BK, K - kk);
        if (clk && tid == 0 && kk / BK < 8) clk[6 + kk / BK] = wall_clock64();      // chunk kk / BK in LDS, next one requested
        hook(sA, kk, kend);
        // The k-steps run in groups of GT_KU (fully unrolled: all fragment reads of a group are in flight before its first MFMA;
        // a rolled loop waits out one LDS round trip per MFMA).  The chunk length is rounded up to whole groups: the staging
        // tiles are zero-filled past K by the providers, so the extra steps add zeros.  NTH = 512: each wave group takes half
        // of the groups.
        constexpr int KG = 4 * GT_KU;
        static_assert(BK % KG == 0, "K chunk must hold whole k-step groups");
        const int kend_r = (kend + KG - 1) / KG * KG;
        int kb = 0, ke = kend_r;
        if (NTH == 512) { const int khalf = ((kend_r / KG + 1) >> 1) * KG; if (grp) kb = khalf; else ke = khalf; }
        for (int k0 = kb; k0 < ke; k0 += KG) {
            float af[GT_KU][C::NSUB], bf[GT_KU][C::NSUB];
#pragma unroll
            for (int u = 0; u < GT_KU; ++u) {
                const int k = k0 + 4 * u;
#pragma unroll
                for (int q = 0; q < C::NSUB; ++q) {
                    const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
                    af[u][q] = AKM ? sA[(k + lg) * C::LDA + ms * 16 + li] : sA[(ms * 16 + li) * C::LDA + k + lg];
                    bf[u][q] = BNK ? sB[(ns * 16 + li) * C::LDB + k + lg] : sB[(k + lg) * C::LDB + ns * 16 + li];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads above ahead of the MFMAs (the scheduler would sink them back)
#pragma unroll
            for (int u = 0; u < GT_KU; ++u) {
#pragma unroll
                for (int q = 0; q < C::NSUB; ++q) acc[q] = mfma16(af[u][q], bf[u][q], acc[q]);
            }
        }
        if (more) {
            __syncthreads();
            stage_commit<C::A_ROWS, C::A_COLS, C::LDA>(sA, ra, tid, kk + BK, afix);
            stage_commit<C::B_ROWS, C::B_COLS, C::LDB>(sB, rb, tid, kk + BK, bfix);
            __syncthreads();
        }
    }
    if (clk && tid == 0) clk[2] = wall_clock64();      // MFMA loop done
    if (NTH == 512) {       // group 1 hands its partial accumulators to group 0 through LDS (operand tiles are dead)
        __syncthreads();
        f32x4* sR = reinterpret_cast<f32x4*>(smem);
        if (grp) {
#pragma unroll
            for (int q = 0; q < C::NSUB; ++q) sR[(wid * C::NSUB + q) * 64 + lane] = acc[q];
        }
        __syncthreads();
        if (grp) return;
#pragma unroll
        for (int q = 0; q < C::NSUB; ++q) {
            const f32x4 o = sR[(wid * C::NSUB + q) * 64 + lane];
            acc[q][0] += o[0]; acc[q][1] += o[1]; acc[q][2] += o[2]; acc[q][3] += o[3];
        }
    }
#pragma unroll
    for (int q = 0; q < C::NSUB; ++q) {
        const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
        const int n = n0 + ns * 16 + li;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) epi(m0 + ms * 16 + 4 * lg + rg, n, acc[q][rg], pf[q][rg]);
    }
    if (clk && tid == 0) clk[3] = wall_clock64();      // epilogue issued
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_tile2: the tile for the MID-SIZE GEMMs (hundreds of MFLOP: scoring at thousands of negatives, wide layers).  Both
// operands K-contiguous in memory (A [m][k], B [n][k]), K a multiple of 32.
//   * 64 x 64 tile, 4 waves as 2 x 2, each wave a 32 x 32 quadrant = ONE v_mfma_f32_32x32x2_f32 accumulator (g4r_device.cuh: the
//     32x32x2 shape sustains 155 TFLOP/s, the 16x16x4 shape 100-126): one A and one B fragment read per MFMA;
//   * K in chunks of 32 through a DOUBLE-buffered LDS tile: the loads of chunk i + 1 are issued before the MFMAs of chunk i and
//     written to the other buffer behind them -- ONE barrier per chunk;
//   * the LDS tile is exactly 32 KiB (FIVE workgroups per CU: at B = 512, N = 8704 all 1088 tiles are resident at once; with
//     padded rows, 35 KB, four fit and the 64 tiles of the second round stretched the launch from 23 to 33 us): rows are 32 floats,
//     element (row, k) sits at k ^ (2 * ((row >> 1) & 15)), which keeps the fragment reads conflict-free (lanes 0-31 = rows of one
//     quadrant, same k: 32 distinct even banks; lanes 32-63 the odd ones) and the staging stores 8-byte aligned;
//   * operands come as ROW POINTERS (`arow(r)` / `brow(r)`: start of row r of the tile, or nullptr for a row outside the
//     matrix / an inactive gathered item), resolved once per thread before the K loop: per chunk a thread spends one 64-bit add
//     per load.  (Counters at B = 512, N = 8704, D = 256, rocprofv3 --pmc: with providers that rebuild (row * D + k) and its
//     clamp for every chunk the VALU work per wave equalled the MFMA work, MFMA pipe 28 % busy.)
//   * epilogue operands (`pre`, 16 per lane) are requested before the K loop.
template <int BK2>
struct Tile2Cfg {
    static constexpr int BM = 64, BN = 64, BK = BK2, LDK = BK2;
    static constexpr int SMEM_FLOATS = 2 * (BM + BN) * LDK;
};
// PRE_COL: `pre` depends on the column only (a bias per column): one fetch per lane instead of one per element.
// BK2 = 32: 32 KiB of LDS per workgroup; BK2 = 16: 16 KiB (the runtime reports 5 workgroups per CU for 32 KiB, but only 4 run:
// with 16 KiB the register file sets the limit, 5).
template <int BK2, bool PRE_COL, class ARow, class BRow, class Pre, class Epi>
__device__ __forceinline__ void gemm_tile2(int m0, int n0, int K, ARow arow, BRow brow, const GAS float* zrow, Pre pre, Epi epi, float* smem, GAS long long* trc = nullptr) {
    using C = Tile2Cfg<BK2>;
    static_assert(BK2 == 16 || BK2 == 32, "chunk depth");
    constexpr int LDK = C::LDK, BUF = 64 * LDK;
    constexpr int TPR = BK2 / 4, RPP = 256 / TPR, NQ = 64 / RPP;      // threads per row, rows per pass, passes (quads per thread and operand)
    constexpr int SWS = BK2 == 32 ? 1 : 2, SWM = BK2 == 32 ? 15 : 7;  // swizzle = 2 * ((row >> SWS) & SWM)
    // buffer b of A at smem + b * BUF, of B at smem + (2 + b) * BUF -- as OFFSETS from the one LDS base: an array of two pointers
    // indexed by (i & 1) loses the address space and the fragment reads compile to flat_load (which also drain vmcnt)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int sr = tid / TPR, sc = 4 * (tid % TPR);      // staging slots of a thread: rows sr + RPP q, k offset sc, of A and of B
    const GAS float* pa[NQ];
    const GAS float* pb[NQ];
    bool oka[NQ], okb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const GAS float* a = arow(sr + RPP * q);
        const GAS float* b = brow(sr + RPP * q);
        oka[q] = a != nullptr; okb[q] = b != nullptr;
        pa[q] = (oka[q] ? a : zrow) + sc;             // rows outside the matrix / inactive gathered rows walk the zero row (>= K floats);
        pb[q] = (okb[q] ? b : zrow) + sc;             // row 0 of the tile is no substitute: it may be inactive itself (null)
    }
    float4 ra[NQ], rb[NQ];
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ra[q] = ld4(pa[q]); rb[q] = ld4(pb[q]); pa[q] += C::BK; pb[q] += C::BK; }
    };
    auto commit = [&](int buf) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 va = oka[q] ? ra[q] : z, vb = okb[q] ? rb[q] : z;
            // swizzle of the row: the quad moves as a whole by bits >= 2, its halves swap by bit 1
            const int row = sr + RPP * q, sw = 2 * ((row >> SWS) & SWM);
            float2* da = reinterpret_cast<float2*>(smem + buf * BUF + row * LDK + (sc ^ (sw & ~3)));
            da[(sw >> 1) & 1] = make_float2(va.x, va.y); da[((sw >> 1) & 1) ^ 1] = make_float2(va.z, va.w);
            float2* db = reinterpret_cast<float2*>(smem + (2 + buf) * BUF + row * LDK + (sc ^ (sw & ~3)));
            db[(sw >> 1) & 1] = make_float2(vb.x, vb.y); db[((sw >> 1) & 1) ^ 1] = make_float2(vb.z, vb.w);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int l32 = lane & 31, lh = lane >> 5;
    const int sw_fr = 2 * ((l32 >> SWS) & SWM);        // swizzle of this lane's fragment row (the same for A and B quadrant rows)
    const int nchunk = K / C::BK;
    issue();
    // epilogue operands of the 16 elements this lane finishes (rows 8 g + 4 lh + r of the quadrant, column l32): requested here,
    // they land during the K loop (fetched behind it, sub-tile by sub-tile, they cost 7 us per tile in the in-kernel trace:
    // all tiles of the launch reach their epilogue together, so nothing else fills the MFMA pipe meanwhile)
    const int n = n0 + wn * 32 + l32;
    constexpr int NPF = PRE_COL ? 1 : 16;
    float4 pf[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) pf[j] = pre(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n);
    commit(0);
    if (trc && tid == 0) trc[2] = wall_clock64();          // first chunk landed and written
    for (int i = 0; i < nchunk; ++i) {
        __syncthreads();                                   // chunk i is in buffer i & 1; buffer (i + 1) & 1 is no longer read
        const float* fa = smem + (i & 1) * BUF + (wm * 32 + l32) * LDK + lh;
        const float* fb = smem + (2 + (i & 1)) * BUF + (wn * 32 + l32) * LDK + lh;
        // groups of 8 k-steps (fragment registers for 8: 5 waves per SIMD need <= 96 registers)
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = fa[(2 * u) ^ sw_fr]; bv[u] = fb[(2 * u) ^ sw_fr]; }
        // chunk i + 1: loads out now (behind the LDS address arithmetic), landed by the end of this chunk's MFMAs, written to the
        // other buffer before the next barrier.  Issue and commit sit in the SAME iteration on purpose: carried around the loop
        // edge, the loaded registers were copied out right behind the loads (a full memory round trip per chunk).
        const bool more = i + 1 < nchunk;
        if (more) issue();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
#pragma unroll
        for (int g = 1; g < BK2 / 16; ++g) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = fa[(16 * g + 2 * u) ^ sw_fr]; bv[u] = fb[(16 * g + 2 * u) ^ sw_fr]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) commit((i + 1) & 1);
    }
    if (trc && tid == 0) trc[3] = wall_clock64();          // K loop done
    // lane holds rows 8 (reg >> 2) + 4 lh + (reg & 3), column l32 of the wave's quadrant: 32 lanes x 4 bytes = 128-byte runs
#pragma unroll
    for (int j = 0; j < 16; ++j) epi(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n, acc[j], pf[PRE_COL ? 0 : j]);
    if (trc && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trc[4] = wall_clock64(); }
}

struct NoFix2k { __device__ __forceinline__ float4 operator()(int, float4 v, bool) const { return v; } };      // gemm_tile2k_full: no post-processing of staged A quads

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_tile3: gemm_tile2's tile (64 x 64, 4 waves 2 x 2, one v_mfma_f32_32x32x2_f32 accumulator per wave, both operands
// K-contiguous) fed by LDS-DMA (`global_load_lds_dwordx4`: global -> LDS without passing through registers) through a ring of
// NST stages of 32 k each.  What it changes against gemm_tile2:
//   * NST - 1 stages are in flight per workgroup (gemm_tile2: one chunk, held in registers), no ds_write pass, ONE barrier per
//     stage; the wait for a stage is a counted `s_waitcnt vmcnt(N)` (hipcc does not see the asm loads, so it neither drains them
//     at the barrier nor waits for them: the waits below are the only ones);
//   * the DMA writes LDS lane-linearly (wave-uniform base + 16 bytes x lane), so a stage is plain [64 rows][8 quads] per operand and
//     the bank spreading is done on the SOURCE side: slot q of row r holds the row's quad q ^ ((r >> 1) & 7); the same involution
//     on the read side makes the 16 lanes of every ds_read_b128 lane group fall on 16 different 16-byte bank columns;
//   * fragments are read as quads (ds_read_b128: one read per operand and FOUR MFMAs): lanes 0-31 take quad 2 j, lanes 32-63 quad
//     2 j + 1 of k-group j, and MFMA u of the group multiplies component u -- a permutation of k inside the group that A and B share;
//   * rows outside the matrix / inactive gathered rows read `zrow` (>= K zero floats in global memory) instead of being masked.
// LDS: NST stages of 2 * 64 * BKS floats.  K must be a multiple of BKS.
template <int NST, int BKS = 32>
struct Tile3Cfg { static constexpr int STAGE = 2 * 64 * BKS, SMEM_FLOATS = NST * STAGE; };

__device__ __forceinline__ void glds16(const GAS float* src, unsigned lds_byte) {      // 64 lanes x 16 bytes -> LDS [lds_byte, + 1 KiB)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}
// Cooperative LDS-DMA copy of a [nrows][nq quads] matrix into an LDS tile whose rows are LDQ quads apart (LDQ >= nq: padded rows
// keep the conflict-free strides of the register-staged layouts).  A piece is 64 / LDQ whole rows; the lanes of pad quads and of
// rows past the matrix are switched off (a masked lane writes nothing: tools/probes/glds_probe.hip), the 8 (NW) waves take pieces
// round robin.  `src(row, q)` -> address of quad q of row `row`.  Completion: the caller's own `s_waitcnt vmcnt` + barrier.
template <int LDQ, int NW, class Src>
__device__ __forceinline__ void dma_rows(unsigned lds_byte, int nrows, int nq, int wid, int lane, Src src) {
    constexpr int RPP = 64 / LDQ;
    const int r = lane / LDQ, q = lane - r * LDQ;
    const bool lane_ok = r < RPP && q < nq;
    for (int p = wid; p * RPP < nrows; p += NW) {
        const int row = p * RPP + r;
        if (lane_ok && row < nrows) glds16(src(row, q), lds_byte + (unsigned)p * (RPP * LDQ * 16));
    }
}
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {      // my LDS-DMA pieces except the newest N have landed; then the workgroup meets
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

// BKS = 32: a row of a stage is 8 quads, a DMA piece (1 KiB) 8 rows, 2 pieces per wave and operand, f(row) = (row >> 1) & 7;
// BKS = 16: 4 quads, 16 rows per piece, 1 piece per wave and operand, f(row) = (row >> 2) & 3 (8 KiB stages: more workgroups per CU).
template <int NST, int BKS, bool PRE_COL, class ARow, class BRow, class Pre, class Epi>
__device__ __forceinline__ void gemm_tile3(int m0, int n0, int K, ARow arow, BRow brow, const GAS float* zrow, Pre pre, Epi epi, float* smem,
                                           GAS long long* trc = nullptr) {
    using C = Tile3Cfg<NST, BKS>;
    static_assert(NST >= 3 && NST <= 5, "ring depth");
    static_assert(BKS == 16 || BKS == 32, "stage depth");
    constexpr int QPR = BKS / 4, RPP = 64 / QPR, NP = 64 / RPP / 4;      // quads per row, rows per piece, pieces per wave and operand
    constexpr int FSH = BKS == 32 ? 1 : 2, FMASK = QPR - 1;              // f(row) = (row >> FSH) & FMASK
    constexpr int LPS = 2 * NP;                                          // DMA pieces per wave and stage
    constexpr unsigned BOFF = 64 * BKS * 4;                              // byte offset of the B half of a stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, lh = lane >> 5;
    const GAS float* pa[NP];
    const GAS float* pb[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = RPP * (NP * wid + j) + lane / QPR;
        const int quad = (lane & FMASK) ^ ((row >> FSH) & FMASK);
        const GAS float* a = arow(row);
        const GAS float* b = brow(row);
        pa[j] = (a ? a : zrow) + 4 * quad;
        pb[j] = (b ? b : zrow) + 4 * quad;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const unsigned piece = lds0 + 1024u * (NP * wid);      // this wave's first A piece inside a stage
    auto issue = [&](int buf) {
        const unsigned base = piece + (unsigned)buf * (C::STAGE * 4);
#pragma unroll
        for (int j = 0; j < NP; ++j) glds16(pa[j], base + 1024u * j);
#pragma unroll
        for (int j = 0; j < NP; ++j) glds16(pb[j], base + BOFF + 1024u * j);
#pragma unroll
        for (int j = 0; j < NP; ++j) { pa[j] += BKS; pb[j] += BKS; }
    };
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int nchunk = K / BKS;
    // epilogue operands first: requested BEFORE the DMA pieces they are older than every piece, so the counted waits below see
    // only pieces behind the stage they wait for
    const int n = n0 + wn * 32 + l32;
    constexpr int NPF = PRE_COL ? 1 : 16;
    float4 pf[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) pf[j] = pre(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) if (s < nchunk) issue(s);
    const int fsw = (l32 >> FSH) & FMASK;                 // f(row) of this lane's fragment rows (A and B: 32 | row offset)
    const float* fa0 = smem + (wm * 32 + l32) * BKS;
    const float* fb0 = smem + 64 * BKS + (wn * 32 + l32) * BKS;
    int buf = 0, nbuf = NST - 1;                          // stage i lives in buffer i % NST; the one issued in iteration i in (i + NST - 1) % NST
    for (int i = 0; i < nchunk; ++i) {
        // stages i + 1 .. i + NST - 2 may stay in flight (LPS pieces each); at the tail fewer are behind stage i
        const int behind = min(NST - 2, nchunk - 1 - i);
        if (NST >= 5 && behind == 3) wait_vm_barrier<3 * LPS>();
        else if (NST >= 4 && behind == 2) wait_vm_barrier<2 * LPS>();
        else if (behind == 1) wait_vm_barrier<LPS>();
        else wait_vm_barrier<0>();
        if (trc && tid == 0 && i == 0) trc[2] = wall_clock64();          // first stage landed
        if (i + NST - 1 < nchunk) issue(nbuf);            // into the buffer stage i - 1 was read from (everyone is past the barrier)
        const float* fa = fa0 + buf * C::STAGE;
        const float* fb = fb0 + buf * C::STAGE;
        constexpr int NG = BKS / 8;                       // k-groups of 8: lanes 0-31 read quad 2 j, lanes 32-63 quad 2 j + 1
        float4 qa[NG], qb[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            qa[j] = *reinterpret_cast<const float4*>(fa + 4 * ((2 * j + lh) ^ fsw));
            qb[j] = *reinterpret_cast<const float4*>(fb + 4 * ((2 * j + lh) ^ fsw));
        }
        __builtin_amdgcn_sched_barrier(0);      // all reads of the stage out before its first MFMA (else: read, wait, 4 MFMAs, read ...)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            acc = mfma32(qa[j].x, qb[j].x, acc);
            acc = mfma32(qa[j].y, qb[j].y, acc);
            acc = mfma32(qa[j].z, qb[j].z, acc);
            acc = mfma32(qa[j].w, qb[j].w, acc);
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
    }
    if (trc && tid == 0) trc[3] = wall_clock64();
#pragma unroll
    for (int j = 0; j < 16; ++j) epi(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n, acc[j], pf[PRE_COL ? 0 : j]);
    if (trc && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trc[4] = wall_clock64(); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_tile2 for operands that are K-MAJOR in memory ([k][m]: a row per k, the 64 elements of the tile contiguous in it) -- the
// GEMMs that contract over the batch (dS = ds^T h) or over gathered rows (dh = ds Sy).  Same tile, MFMA shape and pipeline as
// gemm_tile2; differences:
//   * a K-major operand is staged as [k][64] (one float4 per thread and chunk: thread t -> k row t >> 4, columns 4 (t & 15)), with
//     the two 32-column halves of odd k rows swapped, so that the fragment reads of lanes 0-31 (k even) and 32-63 (k odd) fall on
//     different banks; fragment addresses are lane constants + 128 u floats (immediate offsets);
//   * its provider is called per chunk, `ptr(kk, kr, c)` -> address of element (k = kk + kr, column c) or nullptr (outside the
//     matrix / inactive gathered row): rows of a gathered operand change from chunk to chunk.
// A_KM = false: A is K-contiguous ([m][k], row pointers `arow(r)` as in gemm_tile2, XOR-swizzled rows of 16 floats).
// `safe`: >= 16 readable bytes (DevModel::zrow): what a masked slot loads instead of its operand.  (Round 2 re-read the provider's
// element (0, 0, 0) there, which is itself a null pointer when the FIRST column of a dh slab is an inactive in-batch column -- a slab
// boundary inside [M, B) at the tail of an epoch: a GPU fault at address 0 once B = 240 took these tiles.)
template <bool A_KM, bool PRE_COL, class AProv, class BProv, class Pre, class Epi>
__device__ __forceinline__ void gemm_tile2k(int m0, int n0, int K, AProv aprov, BProv bprov, const GAS float* safe, Pre pre, Epi epi, float* smem, GAS long long* trc = nullptr) {
    constexpr int BK = 16;
    constexpr int BUF = 64 * BK;                        // floats per operand buffer, either layout
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, lh = lane >> 5;
    // K-major staging slot: k row kr, columns kc .. kc + 3
    const int kr = tid >> 4, kc = 4 * (tid & 15);
    const int kofs = kr * 64 + (kc ^ (32 * (kr & 1)));
    // K-contiguous staging slot of A (A_KM == false): row sr, k offset sc (one quad per thread: 64 rows x 16 floats)
    const int sr = tid >> 2, sc = 4 * (tid & 3);
    const int sw_st = 2 * ((sr >> 2) & 7);
    const GAS float* pa = nullptr;
    bool oka = false;
    if constexpr (!A_KM) {
        const GAS float* a = aprov(sr);
        oka = a != nullptr;
        pa = oka ? a + sc : safe;
    }
    // (measured and rejected: (a) an LDS-DMA ring like gemm_tile3's for these operands (3 stages of 32 k: K-major stages need no
    // swizzle, the K-contiguous A is read as quads and the K-major B follows its k order) -- results identical, k_score_bwd2 at
    // B = 512, N = 8704, D = 256 64.6 vs 64.2 us, with 1024 tiles 59.5 vs 60.6: each role alone is bound by how evenly its 64 x 64 x 512
    // tiles (7.8 us of MFMA each) spread over the CUs, not by the staging; (b) three register sets with the loads of chunk i + 2 issued in iteration i -- the compiler's waitcnt
    // placement still drains to the newest load before every commit, and the extra registers cost a workgroup per CU: 68.5 vs 64.5 us)
    // Operand chunks travel global -> registers -> LDS with the loads TWO chunks ahead of the MFMAs (round 3): two register sets, the
    // loads written as asm so that the waits are ours -- `s_waitcnt vmcnt(2)` leaves the newest chunk's two loads in flight while the
    // older chunk is written to LDS (loads return in order).  With compiler-issued loads every commit drained to the newest load
    // (round 2, (b) above), i.e. the gathered rows of the dh slabs arrived one 16-deep chunk (~1 us of MFMAs) ahead of use, less than
    // their latency.  The sets are addressed by compile-time constants (the K loop is unrolled by two): no register indexing.
    // What keeps this safe (guide section 5.7, item 1): hipcc regards an asm load's destination as written when the statement ends,
    // so a register copy it places between the load and our wait reads the register BEFORE the data lands.  The first version of this
    // pipeline waited with `"+v"` operands in two branches (vmcnt(2) / vmcnt(0) at the tail); the allocator merged the branches'
    // outputs with v_mov copies placed in FRONT of the tail's wait: the last chunk of a tile was committed from stale registers
    // whenever its loads took longer than 1.5 iterations -- never on cache-resident tables, a few times per launch on a 10 GB table
    // under load (caught by the catalogue-scale parity test; the ISA audit in tests/test_isa_audit.py now checks the pattern).
    // Now: every wait is a bare `s_waitcnt vmcnt(2)` with no operands and no branch (the tail iterations issue two dummy loads of
    // the zero row so that the count stays 2), followed by a sched_barrier; nothing ties the data registers to the wait, and the
    // commit reads the load's own destination registers.
    f32x4 ra0, rb0, ra1, rb1;
    bool oa0 = false, ob0 = false, oa1 = false, ob1 = false;
    auto ldasm = [](const GAS float* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; };
    auto issue = [&](int kk, f32x4& ra, f32x4& rb, bool& oa, bool& ob) {
        const bool live = kk < K;                       // uniform; past the end: two loads of the zero row, never committed
        if constexpr (A_KM) { const GAS float* a_ = live ? aprov(kk, kr, kc) : nullptr; oa = a_ != nullptr; ra = ldasm(a_ ? a_ : safe); }
        else { oa = oka && live; ra = ldasm(oa ? pa : safe); if (oa) pa += BK; }
        const GAS float* b_ = live ? bprov(kk, kr, kc) : nullptr;
        ob = b_ != nullptr;
        rb = ldasm(b_ ? b_ : safe);
    };
    auto commit = [&](int buf, const f32x4& ra, const f32x4& rb, bool oa, bool ob) {
        const float4 va = make_float4(oa ? ra[0] : 0.f, oa ? ra[1] : 0.f, oa ? ra[2] : 0.f, oa ? ra[3] : 0.f);
        const float4 vb = make_float4(ob ? rb[0] : 0.f, ob ? rb[1] : 0.f, ob ? rb[2] : 0.f, ob ? rb[3] : 0.f);
        if constexpr (A_KM) {
            *reinterpret_cast<float4*>(smem + buf * BUF + kofs) = va;
        } else {
            float2* da = reinterpret_cast<float2*>(smem + buf * BUF + sr * BK + (sc ^ (sw_st & ~3)));
            da[(sw_st >> 1) & 1] = make_float2(va.x, va.y); da[((sw_st >> 1) & 1) ^ 1] = make_float2(va.z, va.w);
        }
        *reinterpret_cast<float4*>(smem + (2 + buf) * BUF + kofs) = vb;
    };
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int sw_fr = 2 * ((l32 >> 2) & 7);             // K-contiguous A: swizzle of this lane's fragment row
    const int nchunk = (K + BK - 1) / BK;
    issue(0, ra0, rb0, oa0, ob0);
#if defined(G4R_MUTATE) && G4R_MUTATE == 4
    // Mutation build 4 (tests only, gru4rec_amd/build.py builds it with the audit switched off): round 3's first version of this
    // pipeline -- waits with tied "+v" operands in two branches, no dummy loads at the tail.  hipcc merges the branches' outputs
    // with copies in FRONT of the tail's wait: the last chunk is committed from registers whose loads may not have landed.
    // The ISA audit must flag it (tests/test_isa_audit.py) and the GPU stress test must turn red on it (tests/test_gpu_stress.py).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra0), "+v"(rb0) :: "memory");
    if (nchunk > 1) issue(BK, ra1, rb1, oa1, ob1);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // chunk 0 (its latency is exposed either way)
    __builtin_amdgcn_sched_barrier(0);
    issue(BK, ra1, rb1, oa1, ob1);
#endif
    const int n = n0 + wn * 32 + l32;
    constexpr int NPF = PRE_COL ? 1 : 16;
    float4 pf[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) pf[j] = pre(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n);
    commit(0, ra0, rb0, oa0, ob0);
    if (trc && tid == 0) trc[2] = wall_clock64();
    // one K chunk out of LDS buffer `buf`: fragments, the loads of the chunk after next, the MFMAs, then the wait for the NEXT chunk
    // (the newest two loads stay in flight; loads return in order, and the epilogue operands requested above are older than both)
    auto chunk = [&](int buf, int i, f32x4& ra, f32x4& rb, bool& oa, bool& ob) {
        __syncthreads();
        const float* fa = A_KM ? smem + buf * BUF + lh * 64 + ((wm * 32 + l32) ^ (32 * lh))
                               : smem + buf * BUF + (wm * 32 + l32) * BK + lh;
        const float* fb = smem + (2 + buf) * BUF + lh * 64 + ((wn * 32 + l32) ^ (32 * lh));
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = A_KM ? fa[128 * u] : fa[(2 * u) ^ sw_fr];
            bv[u] = fb[128 * u];
        }
#if defined(G4R_MUTATE) && G4R_MUTATE == 4
        if (i + 2 < nchunk) issue((i + 2) * BK, ra, rb, oa, ob);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
        __builtin_amdgcn_sched_barrier(0);
#else
        issue((i + 2) * BK, ra, rb, oa, ob);            // into the set chunk i was committed from
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    for (int i = 0; i < nchunk; i += 2) {
        chunk(0, i, ra0, rb0, oa0, ob0);
        if (i + 1 >= nchunk) break;
#if defined(G4R_MUTATE) && G4R_MUTATE == 4
        if (i + 2 < nchunk) asm volatile("s_waitcnt vmcnt(2)" : "+v"(ra1), "+v"(rb1) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra1), "+v"(rb1) :: "memory");
#endif
        commit(1, ra1, rb1, oa1, ob1);
        chunk(1, i + 1, ra1, rb1, oa1, ob1);
        if (i + 2 >= nchunk) break;
#if defined(G4R_MUTATE) && G4R_MUTATE == 4
        if (i + 3 < nchunk) asm volatile("s_waitcnt vmcnt(2)" : "+v"(ra0), "+v"(rb0) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra0), "+v"(rb0) :: "memory");
#endif
        commit(0, ra0, rb0, oa0, ob0);
    }
    // the dummy loads of the tail are still in flight and their registers are free as far as the compiler knows: drain before anything reuses them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (trc && tid == 0) trc[3] = wall_clock64();
#pragma unroll
    for (int j = 0; j < 16; ++j) epi(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n, acc[j], pf[PRE_COL ? 0 : j]);
    if (trc && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trc[4] = wall_clock64(); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_tile2k's tile for SHORT K slices (K <= 16 NCH, NCH <= 8): the whole slice of both operands is requested up front -- 2 NCH quads
// per thread, plain loads in chunk order, so the compiler's own counted waits release chunk after chunk as they land -- and then only
// committed chunk by chunk through the double-buffered LDS tile.  What the split-K kernels without an in-launch join want (a slice
// is a bare GEMM, several of its workgroups share a CU): one memory round trip per workgroup instead of one per two chunks -- the
// A rows of GRU phase 1 are gathered rows of a multi-GB table, ~2 us away, and gemm_tile2k's two-chunks-ahead pipeline paid that
// latency every second chunk (k_gru_p1s at D = 512: 19 us).  No asm loads: nothing to audit.
// A K-contiguous (row pointers, `afix(chunk, quad, ok)` on the way to LDS), B K-major (`bprov(kk, kr, kc)` per chunk), as gemm_tile2k<false>.
template <int NCH, class AProv, class BProv, class Epi, class AFix = NoFix2k>
__device__ __forceinline__ void gemm_tile2k_full(int m0, int n0, int K, AProv aprov, BProv bprov, const GAS float* safe, Epi epi, float* smem, AFix afix = AFix()) {
    static_assert(NCH >= 1 && NCH <= 8, "slice length");
    constexpr int BK = 16, BUF = 64 * BK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, l32 = lane & 31, lh = lane >> 5;
    const int kr = tid >> 4, kc = 4 * (tid & 15);                 // K-major staging slot: k row kr, columns kc .. kc + 3
    const int kofs = kr * 64 + (kc ^ (32 * (kr & 1)));
    const int sr = tid >> 2, sc = 4 * (tid & 3);                  // K-contiguous staging slot of A: row sr, k offset sc
    const int sw_st = 2 * ((sr >> 2) & 7);
    const int nchunk = (K + BK - 1) / BK;
    const GAS float* a = aprov(sr);
    const bool oka = a != nullptr;
    const GAS float* pa = oka ? a + sc : safe;
    float4 ra[NCH], rb[NCH];
    bool ob[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool live = i < nchunk;
        ra[i] = ld4((oka && live) ? pa + i * BK : safe);
        const GAS float* b_ = live ? bprov(i * BK, kr, kc) : nullptr;
        ob[i] = b_ != nullptr;
        rb[i] = ld4(b_ ? b_ : safe);
    }
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int sw_fr = 2 * ((l32 >> 2) & 7);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (i < nchunk) {                                         // (workgroup-uniform)
            const int buf = i & 1;
            float4 va = oka ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            va = afix(i, va, oka);
            const float4 vb = ob[i] ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float2* da = reinterpret_cast<float2*>(smem + buf * BUF + sr * BK + (sc ^ (sw_st & ~3)));
            da[(sw_st >> 1) & 1] = make_float2(va.x, va.y); da[((sw_st >> 1) & 1) ^ 1] = make_float2(va.z, va.w);
            *reinterpret_cast<float4*>(smem + (2 + buf) * BUF + kofs) = vb;
            __syncthreads();                                      // chunk i is in buffer i & 1 (and nobody reads the other buffer any more: its readers passed the previous barrier)
            const float* fa = smem + buf * BUF + (wm * 32 + l32) * BK + lh;
            const float* fb = smem + (2 + buf) * BUF + lh * 64 + ((wn * 32 + l32) ^ (32 * lh));
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = fa[(2 * u) ^ sw_fr]; bv[u] = fb[128 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
        }
    }
    const int n = n0 + wn * 32 + l32;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 16; ++j) epi(m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3), n, acc[j], z4);
}
