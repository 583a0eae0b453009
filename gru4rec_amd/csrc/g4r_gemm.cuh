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
template <int ROWS, int COLS>
struct StageRegs { static constexpr int NQ = (ROWS * (COLS / 4) + 255) / 256; float4 v[NQ]; };

template <int ROWS, int COLS, class Load>
__device__ __forceinline__ void stage_issue(StageRegs<ROWS, COLS>& r, int kk, Load load, int tid) {
    constexpr int C4 = COLS / 4, TOTAL = ROWS * C4;
#pragma unroll
    for (int q = 0; q < StageRegs<ROWS, COLS>::NQ; ++q) {
        // no branch around the load (a conditional definition of v[q] would force an s_waitcnt vmcnt(0) per load):
        // threads beyond the tile re-load its last element and simply do not store it
        const int e = min(tid + 256 * q, TOTAL - 1);
        r.v[q] = load(kk, e / C4, 4 * (e % C4));
    }
}
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ void stage_commit(float* s, const StageRegs<ROWS, COLS>& r, int tid) {
    constexpr int C4 = COLS / 4, TOTAL = ROWS * C4;
#pragma unroll
    for (int q = 0; q < StageRegs<ROWS, COLS>::NQ; ++q) {
        const int e = tid + 256 * q;
        if (e < TOTAL) {
            float* d = s + (e / C4) * LD + 4 * (e % C4);
            if constexpr (LD % 4 == 0) {
                *reinterpret_cast<float4*>(d) = r.v[q];
            } else {
                reinterpret_cast<float2*>(d)[0] = make_float2(r.v[q].x, r.v[q].y);
                reinterpret_cast<float2*>(d)[1] = make_float2(r.v[q].z, r.v[q].w);
            }
        }
    }
}

// One workgroup (256 threads = 4 waves) computes the BM x BN tile at (m0, n0) over K.  smem: TileCfg::SMEM_FLOATS.
// pre(m, n) -> float4 fetches whatever the epilogue needs besides the accumulator (bias, gates, optimizer state)
// BEFORE the K loop, so those loads fly together with the operand staging instead of costing a round trip at the end.
struct NoPre { __device__ __forceinline__ float4 operator()(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); } };

template <int BM, int BN, int BK, bool AKM, bool BNK, class ALoad, class BLoad, class Pre, class Epi>
__device__ __forceinline__ void gemm_tile(int m0, int n0, int K, ALoad aload, BLoad bload, Pre pre, Epi epi, float* smem) {
    using C = TileCfg<BM, BN, BK, AKM, BNK>;
    float* sA = smem;
    float* sB = smem + C::A_ROWS * C::LDA;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lg = lane >> 4;
    constexpr int NT = BN / 16;
    f32x4 acc[C::NSUB];
    float4 pf[C::NSUB][4];
    StageRegs<C::A_ROWS, C::A_COLS> ra;
    StageRegs<C::B_ROWS, C::B_COLS> rb;
    // chunk 0 operands first, then the epilogue operands: the (older) operand loads can be waited for with a
    // counted vmcnt while the epilogue loads are still in flight
    stage_issue(ra, 0, aload, tid);
    stage_issue(rb, 0, bload, tid);
#pragma unroll
    for (int q = 0; q < C::NSUB; ++q) {
        acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) pf[q][rg] = pre(m0 + ms * 16 + 4 * lg + rg, n0 + ns * 16 + li);
    }
    stage_commit<C::A_ROWS, C::A_COLS, C::LDA>(sA, ra, tid);
    stage_commit<C::B_ROWS, C::B_COLS, C::LDB>(sB, rb, tid);
    __syncthreads();
    for (int kk = 0; kk < K; kk += BK) {
        const bool more = kk + BK < K;
        if (more) {         // next chunk's loads fly during this chunk's MFMAs
            stage_issue(ra, kk + BK, aload, tid);
            stage_issue(rb, kk + BK, bload, tid);
        }
        const int kend = min(BK, K - kk);
#pragma unroll 4
        for (int k = 0; k < kend; k += 4) {
#pragma unroll
            for (int q = 0; q < C::NSUB; ++q) {
                const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
                const float a = AKM ? sA[(k + lg) * C::LDA + ms * 16 + li] : sA[(ms * 16 + li) * C::LDA + k + lg];
                const float b = BNK ? sB[(ns * 16 + li) * C::LDB + k + lg] : sB[(k + lg) * C::LDB + ns * 16 + li];
                acc[q] = mfma16(a, b, acc[q]);
            }
        }
        if (more) {
            __syncthreads();
            stage_commit<C::A_ROWS, C::A_COLS, C::LDA>(sA, ra, tid);
            stage_commit<C::B_ROWS, C::B_COLS, C::LDB>(sB, rb, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < C::NSUB; ++q) {
        const int s = wid * C::NSUB + q, ms = s / NT, ns = s % NT;
        const int n = n0 + ns * 16 + li;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) epi(m0 + ms * 16 + 4 * lg + rg, n, acc[q][rg], pf[q][rg]);
    }
}
