// g4r_score_mt.cuh -- part of g4r_step_kernels.cuh (included there behind g4r_fwd_kernels.cuh).  Holds the MACRO-TILE scoring forward,
// k_score_mt: the scores of long score rows / big batches (gru4rec.py:490-497) with ONE workgroup per compute unit and every
// compute unit holding the same amount of work.
//
// Why (measured, rounds 2-5): with 64 x 64 tiles B = 512, N = 8704, D = 256 is 1088 tiles = 4.25 per CU.  All
// tiles of the launch start together, run their K loops together (MFMA bound while they are all resident) and reach their epilogue
// together, so the launch pays prologue + K loop + epilogue one after the other, then a second, nearly empty round for the tiles
// that did not fit: 31 us against 14.5 us of MFMA issue.  Here the score matrix is cut into exactly (number of CUs) macro tiles of
// 64 rows x W columns, W = 64 NB + 16 (8704 / 32 = 272 = 4 x 64 + 16):
//   * 4 waves as 2 x 2; a wave owns NB 32 x 32 blocks (v_mfma_f32_32x32x2_f32) of its 32 rows -- one A fragment feeds NB MFMAs --
//     and a 16 x 16 block (v_mfma_f32_16x16x4_f32) of the 64 x 16 strip that makes the column count come out: every SIMD of the
//     chip issues the same NB x K / 2 + K / 4 MFMAs, no split K, no join;
//   * operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no registers) through a ring of NST 16-deep stages
//     (64 + W rows of 64 bytes: 21 KB at W = 272), NST - 2 stages in flight, one barrier per stage; quad slot q of row r holds the
//     row's quad q ^ ((r >> 2) & 3) (spread on the SOURCE side, the DMA writes lane-linearly), which makes the ds_read_b128
//     fragment reads conflict-free;
//   * one wave per SIMD has nobody to hide behind: whatever it issues between two MFMAs -- the stage wait, the barrier, six DMA
//     pieces, twelve fragment reads -- leaves the matrix pipe idle unless an MFMA is executing meanwhile (first version, everything in
//     front of the stage's 36 MFMAs: 2850 clocks per stage against 2200 of MFMA issue).  So a stage's MFMAs run out of a register
//     set filled one stage earlier, and the wait / barrier / DMA / reads of the next stage are dealt out ONE behind each of its MFMAs
//     (sched_barrier fences keep hipcc from regrouping them); the K loop is unrolled by two so that both register sets
//     are addressed statically;
//   * the waits are the kernel's own counted `s_waitcnt vmcnt(n)` (hipcc does not see the DMA pieces): n = pieces of the stages that
//     may stay in flight.  Other vector-memory operations of a wave only make such a wait more conservative (it retires in order),
//     never weaker.
// Rows past M / inactive columns read the zero row; the epilogue adds the column's bias (- logq * log-popularity) and stores
// 128-byte runs.  The side jobs of k_score_fwd (col_item, occ_idx, occ_fl of this step's columns: atomics on a multi-GB table) are
// dealt over the row blocks of a column group and issued BEHIND the K loop: in front of it their acknowledgements (2-3 us, and
// vector-memory operations retire in order) sat in front of the first stage of the workgroups that had them.
// The hot pointers and sizes are kernel arguments (preloaded into scalar registers) and M comes from the copy of the step state
// staged behind cur_in (a vector load issued with the column items): nothing waits for the descriptor.
// -DMT_X_NO_DMA / _NO_WAIT / _NO_READ / _NO_BAR: ablation builds (results garbage) behind the table in profiles/r06_experiments.md #9.
#pragma once

template <int NB, bool STRIP>
struct MtCfg {
    static constexpr int W = 64 * NB + (STRIP ? 16 : 0);        // columns of a macro tile
    static constexpr int NPB = W / 16;                          // DMA pieces (16 rows x 4 quads = 1 KiB) of the B part of a stage
    static constexpr int NPIECE = 4 + NPB;                      // ... of a stage (A: 64 rows)
    static constexpr int MAXPW = (NPIECE + 3) / 4;              // most pieces a wave issues per stage
    static constexpr int STAGE = (64 + W) * 16;                 // floats
    static constexpr int NST = (148 * 1024 / 4) / STAGE > 8 ? 8 : (148 * 1024 / 4) / STAGE;      // <= 148 KB of the CU's 160
    static constexpr int SMEM_FLOATS = NST * STAGE;
    static_assert(NST >= 4, "ring depth");
};

__device__ __forceinline__ void mt_wait_vm(int n) {      // wave-uniform n: at most n vector-memory operations of this wave still in flight
    switch (n) {
#define MT_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        MT_W(0) MT_W(1) MT_W(2) MT_W(3) MT_W(4) MT_W(5) MT_W(6) MT_W(7) MT_W(8) MT_W(9) MT_W(10) MT_W(11) MT_W(12) MT_W(13) MT_W(14) MT_W(15)
        MT_W(16) MT_W(17) MT_W(18) MT_W(19) MT_W(20) MT_W(21) MT_W(22) MT_W(23) MT_W(24) MT_W(25) MT_W(26) MT_W(27) MT_W(28) MT_W(29) MT_W(30)
        MT_W(31) MT_W(32) MT_W(33) MT_W(34) MT_W(35) MT_W(36)
#undef MT_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int NB, bool STRIP>
struct MtFrag { float4 a[2]; float4 b[NB][2]; float4 sa, sb; };

template <int NB, bool STRIP>
__global__ __launch_bounds__(256) void k_score_mt(const int* __restrict__ ccol_, const int* __restrict__ meta_, const float* __restrict__ hsrc_,
                                                  const float* __restrict__ Wy_, const float* __restrict__ zrow_, const DevModel* __restrict__ mp,
                                                  unsigned dimsA /* D | nrb << 16 */, unsigned N_, unsigned ldc_, unsigned B_) {
    using C = MtCfg<NB, STRIP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GAS int* ccol = (const GAS int*)ccol_;
    const GAS int* meta = (const GAS int*)meta_;
    const GAS float *hsrc = (const GAS float*)hsrc_, *Wy = (const GAS float*)Wy_, *zrow = (const GAS float*)zrow_;
    const int D = (int)(dimsA & 0xFFFFu), nrb = (int)(dimsA >> 16), N = (int)N_, ldc = (int)ldc_, B = (int)B_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1, l32 = lane & 31, lh = lane >> 5, li = lane & 15, lg = lane >> 4;
#if defined(G4R_CLK_TRACE)
    GAS long long* trc = (mp->dbgtile && blockIdx.x < 2048) ? mp->dbgtile + 8 * (size_t)(4096 + blockIdx.x) : nullptr;
#else
    GAS long long* trc = nullptr;
#endif
    if (trc && tid == 0) trc[0] = wall_clock64();
    // the row blocks of a column group sit on ONE XCD (they share the group's gathered rows): tile order = row block fastest inside an
    // XCD's contiguous range of tiles
    const int tile = G4R_XCD_TILE(blockIdx.x, gridDim.x);
    const int cg = tile / nrb, rb = tile - cg * nrb;
    const int m0 = rb * 64, n0 = cg * C::W;
    // ---- first loads: the step's M, the column items of the rows this lane feeds to the DMA and of the columns it finishes
    const int M = meta[2];
    constexpr int MAXPB = (C::NPB + 3) / 4;                     // B pieces of a wave: piece indices wid, wid + 4, ... < NPB
    int it_dma[MAXPB];
#pragma unroll
    for (int j = 0; j < MAXPB; ++j) {
        const int n = n0 + 16 * (wid + 4 * j) + (lane >> 2);
        it_dma[j] = ccol[min(n, ldc - 1)];
        if (n >= ldc || wid + 4 * j >= C::NPB) it_dma[j] = -1;
    }
    int it_col[NB + 1];
#pragma unroll
    for (int q = 0; q <= NB; ++q) {
        const int n = q < NB ? n0 + (wn * NB + q) * 32 + l32 : n0 + 64 * NB + li;
        it_col[q] = ccol[min(n, ldc - 1)];
        if (n >= N) it_col[q] = -1;
    }
    // ---- DMA sources: A piece `wid` (rows 16 wid + lane / 4 of the tile), B pieces wid + 4 j; quad slot (lane & 3) of row r takes the
    // row's quad (lane & 3) ^ ((r >> 2) & 3)
    const int prow = lane >> 2;
    const int squad = (lane & 3) ^ ((prow >> 2) & 3);           // (16 | piece base: f(row) depends on the row inside the piece only)
    const GAS float* pa;
    {
        const int row = m0 + 16 * wid + prow;
        pa = (row < M ? hsrc + (size_t)row * D : zrow) + 4 * squad;
    }
    const GAS float* pb[MAXPB];
#pragma unroll
    for (int j = 0; j < MAXPB; ++j) pb[j] = (it_dma[j] >= 0 ? Wy + (size_t)it_dma[j] * D : zrow) + 4 * squad;
    // epilogue operands (bias - logq * log-popularity of the column's item): requested BEFORE the first DMA piece (older than every piece: the counted waits below only ever see pieces behind the stage they wait for)
    const DevModel& m = *mp;
    const float logq = m.logq;
    const GAS float* By = m.By;
    float bias[NB + 1];
#pragma unroll
    for (int q = 0; q <= NB; ++q) {
        const bool ok = it_col[q] >= 0;
        const int n = q < NB ? n0 + (wn * NB + q) * 32 + l32 : n0 + 64 * NB + li;
        float x = ldf_at(By, max(it_col[q], 0), ok);
        const bool lq = ok && logq != 0.f;
        x -= logq * ldf_at(lq ? (n < B ? m.lq_tgt : m.lq_smp) : By, max(it_col[q], 0), lq);
        bias[q] = x;
    }
    if (trc && tid == 0) { trc[1] = wall_clock64(); trc[5] = (long long)(unsigned)meta[3] | ((long long)meta[4] << 32); }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const int npw = (C::NPIECE - wid + 3) >> 2;                 // pieces this wave issues per stage (wave-uniform)
    const bool last_piece = wid + 4 * (MAXPB - 1) < C::NPB;     // this wave has a B piece in the last round (wave-uniform)
    // one DMA piece of the stage that goes to buffer `buf`: p = 0 the A piece, p = 1 .. MAXPB the B pieces
    auto piece = [&](int buf, int p) {
#if defined(MT_X_NO_DMA)
        if (buf >= 0) return;      // (experiment: no DMA at all -- the results are garbage)
#endif
        const unsigned base = lds0 + (unsigned)buf * (C::STAGE * 4) + 1024u * wid;
        if (p == 0) { glds16(pa, base); pa += 16; }
        else if (p < MAXPB || last_piece) { glds16(pb[p - 1], base + 4096u * p); pb[p - 1] += 16; }
    };
    const int nchunk = D >> 4;
#pragma unroll 1
    for (int s = 0; s < C::NST - 1; ++s) {
        if (s < nchunk) {
#pragma unroll
            for (int p = 0; p <= MAXPB; ++p) piece(s, p);
        }
    }
    f32x16 acc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses inside a stage (floats): A rows of the wave's row block, its NB column blocks, its strip block
    const int fsw = (l32 >> 2) & 3, ssw = (li >> 2) & 3;
    const int oa0 = (wm * 32 + l32) * 16 + 4 * (lh ^ fsw), oa1 = (wm * 32 + l32) * 16 + 4 * ((2 + lh) ^ fsw);
    const int ob0 = 1024 + (wn * NB * 32 + l32) * 16 + 4 * (lh ^ fsw), ob1 = 1024 + (wn * NB * 32 + l32) * 16 + 4 * ((2 + lh) ^ fsw);
    const int osa = (16 * wid + li) * 16 + 4 * (lg ^ ssw), osb = 1024 + (64 * NB + li) * 16 + 4 * (lg ^ ssw);
    using Frag = MtFrag<NB, STRIP>;
    // one fragment read of a stage: r = 0 .. RN - 1 in the order the next stage's MFMAs want them
    constexpr int RN = 2 + 2 * NB + (STRIP ? 2 : 0);
    auto read_one = [&](const float* s, Frag& f, int r) {
#if defined(MT_X_NO_READ)
        if (s != smem) return;     // (experiment: fragments of buffer 0 only)
#endif
        if (r == 0) f.a[0] = *reinterpret_cast<const float4*>(s + oa0);
        else if (r <= NB) f.b[r - 1][0] = *reinterpret_cast<const float4*>(s + ob0 + (r - 1) * 512);
        else if (r == NB + 1) f.a[1] = *reinterpret_cast<const float4*>(s + oa1);
        else if (STRIP && r == NB + 2) f.sa = *reinterpret_cast<const float4*>(s + osa);
        else if (STRIP && r == NB + 3) f.sb = *reinterpret_cast<const float4*>(s + osb);
        else { const int q = r - (NB + 2 + (STRIP ? 2 : 0)); f.b[q][1] = *reinterpret_cast<const float4*>(s + ob1 + q * 512); }
    };
    auto comp = [](const float4& v, int u) { return u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w; };
    // MFMA n = 0 .. MN - 1 of a stage: groups of NB (+ 1 strip MFMA behind the even groups); group g: k-group j = g >> 2, component
    // u = g & 3 of every block
    constexpr int GL0 = NB + (STRIP ? 1 : 0), MN = 4 * GL0 + 4 * NB;      // even groups hold GL0 MFMAs, odd groups NB
    auto mf = [&](const Frag& f, int n) {
        const int pr = n / (GL0 + NB), rem = n - pr * (GL0 + NB);          // pair of groups, position inside it
        const int g = 2 * pr + (rem >= GL0 ? 1 : 0), q = rem >= GL0 ? rem - GL0 : rem;
        const int j = g >> 2, u = g & 3;
        if (STRIP && q == NB) sacc = mfma16(comp(f.sa, g >> 1), comp(f.sb, g >> 1), sacc);
        else acc[q] = mfma32(comp(f.a[j], u), comp(f.b[q][j], u), acc[q]);
    };
#define MT_FENCE() __builtin_amdgcn_sched_barrier(0)
    // stage s lives in buffer s % NST.  Iteration i runs the MFMAs of stage i out of one register set; behind its first MFMA:
    // [stage i + 1 landed; barrier], then ONE item behind each of the following MFMAs -- the DMA pieces of stage i + NST - 1 (into the
    // buffer stage i - 1 was read from: its fragment reads were consumed by the MFMAs of iteration i - 1), then the fragment reads of
    // stage i + 1 into the other register set.  STEADY iterations (a DMA still to issue: NST - 3 younger stages stay in flight) wait
    // with a constant; the last NST - 1 iterations count what is left.
    int rbuf = 0, ibuf = C::NST - 1;                            // buffer of stage i + 1 (set below), buffer the next DMA goes to
    constexpr int PW_HI = (C::NPIECE + 3) / 4, PW_LO = C::NPIECE / 4, NHI = C::NPIECE % 4;      // waves < NHI issue PW_HI pieces per stage
    auto steady_wait = [&]() {
        if (NHI != 0 && wid < NHI) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C::NST - 3) * PW_HI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C::NST - 3) * PW_LO) : "memory");
    };
    static_assert(MN >= 2 + MAXPB + 1 + RN, "more items than MFMA slots");
    // `more`: a stage i + 1 exists; `dma`: a stage i + NST - 1 exists (then the wait is the steady one); tail_n: what may stay in flight otherwise
    auto stage = [&](const Frag& cur, Frag& nxt, bool more, bool dma, int tail_n) {
        const float* s = smem;
#pragma unroll
        for (int n = 0; n < MN; ++n) {
            MT_FENCE();
            mf(cur, n);
            MT_FENCE();
            const int it = n - 1;      // item behind MFMA n: -1 the wait + barrier, 0 .. MAXPB the DMA pieces, then the reads
            if (n == 0) {
                if (more) {
#if !defined(MT_X_NO_WAIT)
                    if (dma) steady_wait(); else mt_wait_vm(tail_n);
#endif
#if !defined(MT_X_NO_BAR)
                    asm volatile("s_barrier" ::: "memory");
#endif
                    rbuf = (rbuf + 1 == C::NST) ? 0 : rbuf + 1;
                    s = smem + rbuf * C::STAGE;
                }
            } else if (it <= MAXPB) {
                if (more && dma) {
                    piece(ibuf, it);
                    if (it == MAXPB) ibuf = (ibuf + 1 == C::NST) ? 0 : ibuf + 1;
                }
            } else if (it - (MAXPB + 1) < RN) {
                if (more) read_one(s, nxt, it - (MAXPB + 1));
            }
        }
        MT_FENCE();
    };
    Frag f0, f1;
    mt_wait_vm(min(C::NST - 2, nchunk - 1) * npw);
    asm volatile("s_barrier" ::: "memory");
    long long cyc0 = 0;
    if (trc && tid == 0) { trc[2] = wall_clock64(); cyc0 = clock64(); }
#pragma unroll
    for (int r = 0; r < RN; ++r) read_one(smem, f0, r);
    int i = 0;                                                  // (nchunk is even: D a multiple of 32, host)
    for (; i + C::NST < nchunk; i += 2) { stage(f0, f1, true, true, 0); stage(f1, f0, true, true, 0); }
    for (; i < nchunk; i += 2) {
        stage(f0, f1, true, i + C::NST - 1 < nchunk, (nchunk - 2 - i) * npw);
        stage(f1, f0, i + 2 < nchunk, i + C::NST < nchunk, (nchunk - 3 - i) * npw);
    }
#undef MT_FENCE
    if (trc && tid == 0) { trc[3] = wall_clock64(); trc[7] = clock64() - cyc0; }
    // ---- side jobs of the scoring forward, dealt over the row blocks of the column group: this step's column -> item list, its part
    // of the gathered-row list and the first / last occurrence marks of its items (k_sparse_update)
    {
        const int per = (C::W + nrb - 1) / nrb;
        for (int e = rb * per + tid; e < min((rb + 1) * per, C::W); e += 256) {
            const int n = n0 + e;
            if (n < ldc) {
                const int item = ccol[n];
                m.col_item[n] = item;
                if (n < N) {
                    m.occ_idx[B + n] = item;
                    if (item >= 0 && m.xmode == 0) {
                        int* fl = (int*)m.occ_fl + 4 * (size_t)item;
                        atomicMax(fl, B + n + 1);
                        atomicMax(fl + 1, m.R - (B + n));
                        atomicAdd(fl + 2, 1);
                    }
                }
            }
        }
    }
    // ---- epilogue: lane holds rows 8 (j >> 2) + 4 lh + (j & 3), column l32 of each 32 x 32 block; rows 4 lg + j, column li of its strip block
    GAS float* Sc = m.Sc;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int n = n0 + (wn * NB + q) * 32 + l32;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = m0 + wm * 32 + 8 * (j >> 2) + 4 * lh + (j & 3);
            if (row < M && n < N) Sc[(size_t)row * ldc + n] = acc[q][j] + bias[q];
        }
    }
    if constexpr (STRIP) {
        const int n = n0 + 64 * NB + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = m0 + 16 * wid + 4 * lg + j;
            if (row < M && n < N) Sc[(size_t)row * ldc + n] = sacc[j] + bias[NB];
        }
    }
    if (trc && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trc[4] = wall_clock64();
#if defined(G4R_CLK_TRACE)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[6] = (long long)hw | ((long long)(xcc & 0xF) << 32);
#endif
    }
}
template __global__ void k_score_mt<4, true>(const int* __restrict__, const int* __restrict__, const float* __restrict__, const float* __restrict__,
                                             const float* __restrict__, const DevModel* __restrict__, unsigned, unsigned, unsigned, unsigned);
