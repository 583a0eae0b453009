// g4r_score_bmt.cuh -- part of g4r_step_kernels.cuh (included there behind g4r_bwd_kernels.cuh).  Holds the MACRO-TILE scoring backward,
// k_score_bmt (gru4rec.py:383-384 through the scores of :490-497): both products of the scoring backward at long score rows / big
// batches cut into 2 x (number of CUs) workgroups of EQUAL MFMA work, two per compute unit:
//   role A  dS[n][d] = sum_b ds[b][n] h[b][d]   (N x D, K = batch): 272 x 32 tiles (N / 272 column groups x D / 32); a wave owns two
//           32 x 32 blocks (rows 2 i + u of its 64: ONE ds_read_b64 feeds both) and a K half of a 16 x 16 block of the 16-row strip;
//           the tile stores RAW gradient rows into the step plane dSy; the Adagrad rule of k_score_bwd2's role A (step rows, accumulators in
//           place for single-occurrence items, dAy otherwise) needs the accumulator rows of the step's items -- a gather that every tile of
//           the chip would pay at the same moment (9-11 us in the epilogue, 13 us in front of the first stage: vector memory retires in order,
//           so the counted waits of the K loop cannot leave it in flight) -- and runs as extra workgroups of k_gru_bwd_a, the launch two
//           behind this one, which leaves half of the chip idle (score_fin_rows, g4r_bwd_kernels.cuh)
//   role B  dh[b][d] = sum_n ds[b][n] Wy[item_n][d]  (B x D, K = score columns): 64 x 128 tiles x ldSc / kch slabs of kch columns,
//           written as split-K planes dhpart[slab] (summed in slab order by the GRU backward); a wave owns 32 rows x two blocks of 32
//           columns (columns 2 j + u of its 64: one ds_read_b64 feeds both, one A fragment feeds both)
// With B = 512, N = 8704, D = 256: 32 x 8 = 256 role-A and 8 x 2 x 16 = 256 role-B workgroups of 34.8 K / 34.8 K clocks of MFMA issue
// each -- k_score_bwd2's 64 x 64 tiles are 4.25 per CU and its two roles do not end together (58 us against 29 us of MFMA issue).
// Two workgroups of <= 78 KB of LDS share a CU: two waves per SIMD with independent barriers fill each other's holes (k_score_mt runs one).
// Both roles: operands global -> LDS by LDS-DMA through a ring of 16-deep stages, the kernel's own counted waits, the MFMAs of a stage
// out of registers filled one stage earlier, wait / barrier / DMA pieces / fragment reads of the next stage dealt between them
// (mt_pipeline; g4r_score_mt.cuh describes the scheme).  K-major operands (ds and h over the batch; the gathered rows) are staged as
// [k][columns]: a DMA piece is 64 consecutive quads of that, fragments are b32 / b64 reads of consecutive columns (conflict-free).
// The bias gradient (k_score_bwd2's role C: column sums of ds over the batch) rides on role A, whose tiles see every batch row of their
// 272 columns: each of the D / 32 tiles of a column group sums 272 / (D / 32) of them out of the staged ds rows (threads < 4 x 34: a column
// and a quarter of a stage's 16 rows each, four LDS reads per stage), and finishes them with the Adagrad rule of the item rows.
// (Measured first as extra workgroups of the element-wise launch behind this one: +8.6 us there.)
#pragma once

// The pipeline of a macro-tile workgroup (4 waves).  NST ring stages of `stage_floats`; per stage and wave: MN MFMAs (mf(frag, n)), NP
// DMA pieces at most (piece(buf, p) issues piece p of the NEXT stage to issue into ring buffer buf -- and skips the pieces this wave
// does not own), RN fragment reads (read_one(stage base, frag, r)).  Pieces per stage and wave: PW_HI for waves < NHI, PW_LO for the
// others (NHI = 0: PW_LO everywhere).  nchunk (stages) must be even.  Item order behind the MFMAs of stage i: [stage i + 1 landed;
// barrier] behind MFMA 0, then the pieces of stage i + NST - 1, then the reads of stage i + 1, IPS items per MFMA slot.
template <int NST, int MN, int NP, int RN, int PW_HI, int PW_LO, int NHI, class Frag, class Piece, class Read, class Mf>
__device__ __forceinline__ void mt_pipeline(int nchunk, int wid, int stage_floats, float* smem, Piece piece, Read read_one, Mf mf, GAS long long* trc) {
    constexpr int NI = NP + RN, IPS = (NI + MN - 2) / (MN - 1);
    static_assert(NST >= 4 && (NST - 2) * PW_HI <= 36, "ring depth / wait table");
    const int npw = (NHI != 0 && wid < NHI) ? PW_HI : PW_LO;
#pragma unroll 1
    for (int s = 0; s < NST - 1; ++s) {
        if (s < nchunk) {
#pragma unroll
            for (int p = 0; p < NP; ++p) piece(s, p);
        }
    }
    int rbuf = 0, ibuf = NST - 1;
    auto steady_wait = [&]() {
        if (NHI != 0 && wid < NHI) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 3) * PW_HI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 3) * PW_LO) : "memory");
    };
    auto stage = [&](const Frag& cur, Frag& nxt, bool more, bool dma, int tail_n) {
        const float* s = smem;
#pragma unroll
        for (int n = 0; n < MN; ++n) {
            __builtin_amdgcn_sched_barrier(0);
            mf(cur, n);
            __builtin_amdgcn_sched_barrier(0);
            if (n == 0) {
                if (more) {
                    if (dma) steady_wait(); else mt_wait_vm(tail_n);
                    asm volatile("s_barrier" ::: "memory");
                    rbuf = (rbuf + 1 == NST) ? 0 : rbuf + 1;
                    s = smem + rbuf * stage_floats;
                }
            } else {
#pragma unroll
                for (int e = 0; e < IPS; ++e) {
                    const int it = (n - 1) * IPS + e;
                    if (it < NP) {
                        if (more && dma) {
                            piece(ibuf, it);
                            if (it == NP - 1) ibuf = (ibuf + 1 == NST) ? 0 : ibuf + 1;
                        }
                    } else if (it < NI) {
                        if (more) read_one(s, nxt, it - NP);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    Frag f0, f1;
    mt_wait_vm(min(NST - 2, nchunk - 1) * npw);
    asm volatile("s_barrier" ::: "memory");
    if (trc && threadIdx.x == 0) trc[2] = wall_clock64();
#pragma unroll
    for (int r = 0; r < RN; ++r) read_one(smem, f0, r);
    int i = 0;
    for (; i + NST < nchunk; i += 2) { stage(f0, f1, true, true, 0); stage(f1, f0, true, true, 0); }
    for (; i < nchunk; i += 2) {
        stage(f0, f1, true, i + NST - 1 < nchunk, (nchunk - 2 - i) * npw);
        stage(f1, f0, i + 2 < nchunk, i + NST < nchunk, (nchunk - 3 - i) * npw);
    }
    if (trc && threadIdx.x == 0) trc[3] = wall_clock64();
}

#define BMT_WA 272                     // score columns of a role-A tile
#define BMT_NST_A 4                    // ring stages of role A: 16 x (272 + 32) floats = 19 KB each
#define BMT_NST_B 6                    // ... of role B: (64 x 16 + 16 x 128) floats = 12 KB each
#define BMT_STAGE_A (16 * (BMT_WA + 32))
#define BMT_STAGE_B (64 * 16 + 16 * 128)
struct BmtFragA { float2 a[8]; float b[8]; float sa[2], sb[2]; float bs[4]; };
struct BmtFragB { float4 a[2]; float2 b[8]; };

// grid: 2 x ntile workgroups; workgroup id -> role ((id >> 3) ^ (id >> 8)) & 1, tile ((id >> 4) << 3) | (id & 7): every XCD (id mod 8) gets both
// roles alternately, and its k-th and (k + 32)-th workgroup (the pair that shares a CU when the dispatcher deals one per CU first) one of each.  Role A tile t (XCD-contiguous order) -> column group t / ndg, d group t % ndg; role B tile t -> slab t / (nrt * 2),
// row tile, d part of 128.  kch = slab depth (a multiple of 32), ntile = number of tiles of EACH role.
__global__ __launch_bounds__(256, 2) void k_score_bmt(const float* __restrict__ ds_, const float* __restrict__ h_, const float* __restrict__ Wy_,
                                                      const int* __restrict__ item_, const int* __restrict__ meta_, const float* __restrict__ zrow_,
                                                      const DevModel* __restrict__ mp, unsigned dimsA /* D | ndg << 16 */, unsigned dimsB /* N | ld << 16 */,
                                                      unsigned dimsC /* B | kch << 16 */, unsigned dimsD /* row tiles of 64 | d parts of 128 << 16 */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GAS float *ds = (const GAS float*)ds_, *h = (const GAS float*)h_, *Wy = (const GAS float*)Wy_, *zrow = (const GAS float*)zrow_;
    const GAS int *colitem = (const GAS int*)item_, *meta = (const GAS int*)meta_;
    const int D = (int)(dimsA & 0xFFFFu), ndg = (int)(dimsA >> 16), N = (int)(dimsB & 0xFFFFu), ld = (int)(dimsB >> 16);
    const int B = (int)(dimsC & 0xFFFFu), kch = (int)(dimsC >> 16), nrt = (int)(dimsD & 0xFFFFu), ndh = (int)(dimsD >> 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, lh = lane >> 5, li = lane & 15, lg = lane >> 4;
    const int role = (((int)blockIdx.x >> 3) ^ ((int)blockIdx.x >> 8)) & 1;      // (an XCD's k-th and (k + 32)-th workgroup -- the two a CU gets -- differ in role)
    const int ntile = (int)gridDim.x >> 1;
    const int tile = G4R_XCD_TILE((((int)blockIdx.x >> 4) << 3) | ((int)blockIdx.x & 7), ntile);
#if defined(G4R_CLK_TRACE)
    GAS long long* trc = (mp->dbgtile && blockIdx.x < 2048) ? mp->dbgtile + 8 * (size_t)(4096 + 2048 + blockIdx.x) : nullptr;
#else
    GAS long long* trc = nullptr;
#endif
    if (trc && tid == 0) { trc[0] = wall_clock64(); trc[6] = role; }
    const int M = meta[2];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    if (role == 0) {
        // ------------------------------------------------------------------------------------------------ role A
        const int cg = tile / ndg, dg = tile - cg * ndg;
        const int n0 = cg * BMT_WA, d0 = dg * 32;
        // DMA pieces: a stage is [16 k][76 quads] (68 of ds, 8 of h) = 19 pieces of 64 consecutive quads; wave w owns pieces w + 4 p
        constexpr int NPA = 5;
        const GAS float* src[NPA];
        int left[NPA];                      // batch rows left below this lane's row of the stage about to be issued (<= 0: zero row)
        unsigned inc[NPA];                  // floats per stage
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int f = 64 * (wid + 4 * p) + lane, k = f / 76, q = f - 76 * k;
            const bool isds = q < 68;
            src[p] = isds ? ds + (size_t)k * ld + n0 + 4 * q : h + (size_t)k * D + d0 + 4 * (q - 68);
            inc[p] = isds ? 16u * (unsigned)ld : 16u * (unsigned)D;
            left[p] = M - k;
        }
        __builtin_amdgcn_s_setprio(2);      // role A ends with its stores, role B with next to nothing: let A's waves of a SIMD reach theirs first
        // bias gradient: this tile's share of the column group's columns, BMT_WA / ndg of them; thread -> column bc, rows 4 bq .. 4 bq + 3 of a stage
        const int nbc = BMT_WA / ndg;                                        // (host: ndg divides 272 ... 8 -> 34)
        const int bq4 = tid / nbc, bc = tid - bq4 * nbc;
        const bool bias_thr = tid < 4 * nbc;
        const int obs = (4 * bq4) * (BMT_WA + 32) + dg * nbc + bc;
        float bsum = 0.f;
        if (trc && tid == 0) { trc[1] = wall_clock64(); trc[5] = (long long)(unsigned)meta[3] | ((long long)meta[4] << 32); }
        auto piece = [&](int buf, int p) {
            if (p < NPA - 1 || wid < 3) {                                   // (19 pieces: wave 3 has four)
                glds16(left[p] > 0 ? src[p] : zrow, lds0 + (unsigned)buf * (BMT_STAGE_A * 4) + 1024u * (wid + 4 * p));
                src[p] += inc[p];
                left[p] -= 16;
            }
        };
        f32x16 acc0, acc1;
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
        f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int sb = wid & 1, kh = wid >> 1;                               // strip: d half, K half
        const int oa = lh * (BMT_WA + 32) + 64 * wid + 2 * l32, ob = lh * (BMT_WA + 32) + BMT_WA + l32;
        const int osa = (8 * kh + lg) * (BMT_WA + 32) + 256 + li, osb = (8 * kh + lg) * (BMT_WA + 32) + BMT_WA + 16 * sb + li;
        auto read_one = [&](const float* s, BmtFragA& f, int r) {
            if (r < 16) {
                const int st = r >> 1;                                       // k-step: k = 2 st + lh
                if ((r & 1) == 0) f.a[st] = *reinterpret_cast<const float2*>(s + oa + 2 * st * (BMT_WA + 32));
                else f.b[st] = s[ob + 2 * st * (BMT_WA + 32)];
            } else if (r < 20) {
                const int t = (r - 16) >> 1;                                 // strip MFMA t: k = 8 kh + 4 t + lg
                if ((r & 1) == 0) f.sa[t] = s[osa + 4 * t * (BMT_WA + 32)];
                else f.sb[t] = s[osb + 4 * t * (BMT_WA + 32)];
            } else {
                f.bs[r - 20] = bias_thr ? s[obs + (r - 20) * (BMT_WA + 32)] : 0.f;      // (rows past M hold zeros: the zero row)
            }
        };
        auto mf = [&](const BmtFragA& f, int n) {      // 18 MFMAs: k-steps 0 .. 3 (two each), strip 0, k-steps 4 .. 7, strip 1
            const int m = n < 8 ? n : (n == 8 ? -1 : (n < 17 ? n - 1 : -2));
            if (n == 4) bsum += (f.bs[0] + f.bs[1]) + (f.bs[2] + f.bs[3]);
            if (m == -1) sacc = mfma16(f.sa[0], f.sb[0], sacc);
            else if (m == -2) sacc = mfma16(f.sa[1], f.sb[1], sacc);
            else if ((m & 1) == 0) acc0 = mfma32(f.a[m >> 1].x, f.b[m >> 1], acc0);
            else acc1 = mfma32(f.a[m >> 1].y, f.b[m >> 1], acc1);
        };
        const int nchunk = ((M + 31) >> 5) << 1;                             // 16-row stages, an even number of them
        mt_pipeline<BMT_NST_A, 18, NPA, 24, 5, 4, 3, BmtFragA>(nchunk, wid, BMT_STAGE_A, smem, piece, read_one, mf, trc);
        // ---- the strip's K halves meet in LDS (the ring is dead behind this barrier)
        __syncthreads();
        f32x4* sred = reinterpret_cast<f32x4*>(smem);
        if (kh == 1) sred[sb * 64 + lane] = sacc;
        float* sbias = smem + 4 * 128;                                       // [4 row quarters][nbc]
        if (bias_thr) sbias[tid] = bsum;
        __syncthreads();
        if (kh == 0) { const f32x4 o = sred[sb * 64 + lane]; sacc[0] += o[0]; sacc[1] += o[1]; sacc[2] += o[2]; sacc[3] += o[3]; }
        // ---- Adagrad epilogue (k_score_bwd2 role A): lane holds rows n = n0 + 64 wid + 2 (8 (j >> 2) + 4 lh + (j & 3)) + u, column d0 + l32
        const DevModel& m = *mp;
        const float lr = m.lr;
        const bool generic = m.generic != 0;
        const GAS int* occ_fl = m.occ_fl;
        const long long g_ = (long long)(unsigned)meta[0] | ((long long)meta[1] << 32);
        GAS float* dSy = G4R_DSY(m, g_);
        {
            // the bias gradient of this tile's columns (threads < nbc): quarters added in row order
            if (tid < nbc) {
                const int n = n0 + dg * nbc + tid;
                const float g = (sbias[tid] + sbias[nbc + tid]) + (sbias[2 * nbc + tid] + sbias[3 * nbc + tid]);
                if (n < N) {
                    const int item = colitem[n];
                    const bool ok = item >= 0;
                    const int c1 = occ_fl[4 * (size_t)max(item, 0) + 2];
                    const float an = ldf_at(m.accBy, max(item, 0), ok) + G4R_MUT_ACC(g * g);
                    float step = ok ? G4R_MUT_STEP(lr * g * frsq(an + G4R_EPS_ADAGRAD)) : 0.f;
                    if (generic) step = ok ? g : 0.f;
                    G4R_DSBY(m, g_)[n] = step;
                    if (!generic && ok && c1 == 1) m.accBy[item] = an; else m.dABy[n] = an;
                }
            }
            // the RAW gradient rows go to the step plane (128-byte runs); score_fin_rows (riding on k_gru_bwd_a) turns them into Adagrad steps
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int n = n0 + 64 * wid + 2 * (8 * (j >> 2) + 4 * lh + (j & 3)) + u;
                    if (n < N) dSy[(size_t)n * D + d0 + l32] = u == 0 ? acc0[j] : acc1[j];
                }
            if (kh == 0) {                                                   // strip rows n0 + 256 + 4 lg + j, column d0 + 16 sb + li
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int n = n0 + 256 + 4 * lg + j; if (n < N) dSy[(size_t)n * D + d0 + 16 * sb + li] = sacc[j]; }
            }
        }
        if (trc && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            trc[4] = wall_clock64();
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            trc[7] = (long long)hw | ((long long)(xcc & 0xF) << 32);
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------------- role B
    {
        const int per_kc = nrt * ndh;
        const int kc = tile / per_kc, rem = tile - kc * per_kc, rt = rem / ndh, dh = rem - rt * ndh;
        const int m0 = rt * 64, d0 = dh * 128, kbeg = kc * kch;
        const int wm = wid >> 1, wn = wid & 1;
        // the slab's column items, read by the DMA lanes stage by stage
        int* sIt = reinterpret_cast<int*>(smem + BMT_NST_B * BMT_STAGE_B);
        for (int i = tid; i < kch; i += 256) sIt[i] = (kbeg + i < ld) ? colitem[kbeg + i] : -1;
        // A piece `wid`: rows 16 wid + lane / 4 of ds (K-contiguous), quad slot (lane & 3) takes quad (lane & 3) ^ ((row >> 2) & 3)
        const int prow = lane >> 2, squad = (lane & 3) ^ ((prow >> 2) & 3);
        const GAS float* pa;
        {
            const int row = m0 + 16 * wid + prow;
            pa = (row < M) ? ds + (size_t)row * ld + kbeg + 4 * squad : zrow + 4 * squad;
        }
        const bool arow_ok = m0 + 16 * wid + prow < M;
        // B pieces wid and wid + 4 of [16 k][32 quads]: lane -> k = 2 piece + (lane >> 5), quad lane & 31
        const int bq = lane & 31;
        int kk[2] = {2 * wid + (lane >> 5), 2 * (wid + 4) + (lane >> 5)};
        __syncthreads();
        int itn[2] = {sIt[kk[0]], sIt[kk[1]]};                               // items of the stage about to be issued
        int sidx = 0;                                                        // ... and its first column inside the slab
        if (trc && tid == 0) { trc[1] = wall_clock64(); trc[5] = (long long)(unsigned)meta[3] | ((long long)meta[4] << 32); }
        auto piece = [&](int buf, int p) {
            const unsigned base = lds0 + (unsigned)buf * (BMT_STAGE_B * 4);
            if (p == 0) {
                glds16(pa, base + 1024u * wid);
                if (arow_ok) pa += 16;
            } else {
                const int it = itn[p - 1];
                glds16(it >= 0 ? Wy + (size_t)it * D + d0 + 4 * bq : zrow, base + 4096u + 1024u * (wid + 4 * (p - 1)));
                itn[p - 1] = sIt[min(sidx + 16 + kk[p - 1], kch - 1)];     // (past the slab: never issued)
                if (p == 2) sidx += 16;
            }
        };
        f32x16 acc0, acc1;
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
        const int fsw = (l32 >> 2) & 3;
        const int oa0 = (wm * 32 + l32) * 16 + 4 * (lh ^ fsw), oa1 = (wm * 32 + l32) * 16 + 4 * ((2 + lh) ^ fsw);
        const int ob = 1024 + 64 * wn + 2 * l32;
        auto read_one = [&](const float* s, BmtFragB& f, int r) {
            if (r == 0) f.a[0] = *reinterpret_cast<const float4*>(s + oa0);
            else if (r == 1) f.a[1] = *reinterpret_cast<const float4*>(s + oa1);
            else {
                const int e = r - 2, j = e >> 2, u = e & 3;                  // MFMA pair (j, u): k = 4 (2 j + lh) + u
                f.b[e] = *reinterpret_cast<const float2*>(s + ob + (4 * (2 * j + lh) + u) * 128);
            }
        };
        auto comp = [](const float4& v, int u) { return u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w; };
        auto mf = [&](const BmtFragB& f, int n) {
            const int e = n >> 1, j = e >> 2, u = e & 3;
            if ((n & 1) == 0) acc0 = mfma32(comp(f.a[j], u), f.b[e].x, acc0);
            else acc1 = mfma32(comp(f.a[j], u), f.b[e].y, acc1);
        };
        mt_pipeline<BMT_NST_B, 16, 3, 10, 3, 3, 0, BmtFragB>(kch >> 4, wid, BMT_STAGE_B, smem, piece, read_one, mf, trc);
        // ---- split-K plane of this slab: rows m0 + 32 wm + 8 (j >> 2) + 4 lh + (j & 3), columns d0 + 64 wn + 2 l32 + {0, 1}
        GAS float* dhpart = mp->dhpart;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = m0 + 32 * wm + 8 * (j >> 2) + 4 * lh + (j & 3);
            if (b < M) *reinterpret_cast<GAS float2*>(dhpart + ((size_t)kc * B + b) * D + d0 + 64 * wn + 2 * l32) = make_float2(acc0[j], acc1[j]);
        }
        if (trc && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            trc[4] = wall_clock64();
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            trc[7] = (long long)hw | ((long long)(xcc & 0xF) << 32);
        }
    }
}

