// Multi-rank reconciliation of the GPU-local item tables (Wy / By / E and their optimizer state).  New: the reference is
// single-GPU; north_star keeps the sparse rows GPU-local between synchronisation points.  For the rows some rank touched since
// the last synchronisation (`base` = the common value at that point) every replica of a table ends at
//     G4R_SYNC_SUM    base + sum  over the ranks q that touched the row, in rank order, of (value_q - base)
//     G4R_SYNC_MEAN   base + mean over the ranks q that touched the row               of (value_q - base)
// Under either rule a row only one rank trained receives exactly that rank's update (averaging whole replicas would keep 1 / nranks
// of it).  SUM keeps every rank's update in full -- right for additive statistics (Adagrad's sum of squared gradients) -- but on
// PARAMETERS it applies N independent full-size steps from the same starting point: measured with virtual ranks (DESIGN.md
// section 7) Recall@20 falls from 0.41 to 0.17 at two ranks and the loss diverges at eight; MEAN is the local-SGD rule.  Rows are
// exchanged as packed (id list, delta rows) parts; the kernels below are shared by the RCCL path (g4r_comm_sync_sparse) and by the
// host-driven test hooks (g4r_sync_export / g4r_sync_import).
#pragma once
#include "g4r_device.cuh"

// out[j][c] = cur[ids[j]][c] - base[ids[j]][c]
__global__ __launch_bounds__(256) void k_sync_pack(const float* cur, const float* base, int W, const int* ids, long long n, float* out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * W) return;
    const long long j = e / W;
    const int c = (int)(e - j * W);
    const size_t o = (size_t)ids[j] * W + c;
    out[e] = cur[o] - base[o];
}
// cur[ids[j]] = base[ids[j]]
__global__ __launch_bounds__(256) void k_sync_reset(float* cur, const float* base, int W, const int* ids, long long n) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * W) return;
    const long long j = e / W;
    const size_t o = (size_t)ids[j] * W + (int)(e - j * W);
    cur[o] = base[o];
}
// cur[ids[j]] += delta[j] (rowcnt == nullptr: SUM) or delta[j] / (number of parts that hold the row) (MEAN)
// (ids of one part are distinct: no two threads meet on an element)
__global__ __launch_bounds__(256) void k_sync_add(float* cur, int W, const int* ids, long long n, const float* delta, const unsigned char* rowcnt) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * W) return;
    const long long j = e / W;
    const size_t o = (size_t)ids[j] * W + (int)(e - j * W);
    float dlt = delta[e];
    if (rowcnt) { const int c = rowcnt[ids[j]]; if (c > 1) dlt = dlt / (float)c; }
    cur[o] += dlt;
}
// rowcnt[ids[j]] += 1 (one launch per part, in stream order: the ids of a part are distinct) / = 0
__global__ __launch_bounds__(256) void k_sync_count(unsigned char* rowcnt, const int* ids, long long n, int clear) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    rowcnt[ids[j]] = clear ? 0 : (unsigned char)(rowcnt[ids[j]] + 1);
}
// base[ids[j]] = cur[ids[j]]
__global__ __launch_bounds__(256) void k_sync_rebase(const float* cur, float* base, int W, const int* ids, long long n) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * W) return;
    const long long j = e / W;
    const size_t o = (size_t)ids[j] * W + (int)(e - j * W);
    base[o] = cur[o];
}

// ---- the touched bitmap -> a sorted id list, on the device (round 4: the 10 M-item bitmap used to be copied to the host and walked
// there, 5-8 ms of a 10.7 ms reconciliation at the configs[3] shape).  TC_CHUNK bytes per workgroup: count, exclusive scan of the
// workgroup counts (one workgroup, running carry), then every workgroup writes its ids behind its offset, in item order.
#define TC_PER_THREAD 16
#define TC_CHUNK (256 * TC_PER_THREAD)
__global__ __launch_bounds__(256) void k_touched_count(const unsigned char* t, long long I, int* blockcnt) {
    __shared__ int red[4];
    const long long i0 = (long long)blockIdx.x * TC_CHUNK + (long long)threadIdx.x * TC_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int e = 0; e < TC_PER_THREAD; ++e) c += (i0 + e < I && t[i0 + e]) ? 1 : 0;
    c = (int)wave_sum((float)c);      // (<= 1024 per wave: exact in fp32)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// offsets[b] = sum of blockcnt[0 .. b), offsets[nb] = total
__global__ __launch_bounds__(1024) void k_touched_scan(const int* blockcnt, int nb, int* offsets) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? blockcnt[b] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {      // inclusive scan (Hillis-Steele)
            const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (b < nb) offsets[b] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[nb] = carry;
}
__global__ __launch_bounds__(256) void k_touched_write(const unsigned char* t, long long I, const int* offsets, int* ids) {
    __shared__ int pre[256];
    const long long i0 = (long long)blockIdx.x * TC_CHUNK + (long long)threadIdx.x * TC_PER_THREAD;
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < TC_PER_THREAD; ++e) bits |= (i0 + e < I && t[i0 + e]) ? (1u << e) : 0u;
    const int c = __popc(bits);
    pre[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int add = threadIdx.x >= d ? pre[threadIdx.x - d] : 0;
        __syncthreads();
        pre[threadIdx.x] += add;
        __syncthreads();
    }
    int o = offsets[blockIdx.x] + pre[threadIdx.x] - c;
#pragma unroll
    for (int e = 0; e < TC_PER_THREAD; ++e) if (bits & (1u << e)) ids[o++] = (int)(i0 + e);
}
__global__ __launch_bounds__(256) void k_fill_i32(int* p, long long n, int v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- dense form of the reconciliation (item tables of up to a few tens of MB: all device side, no host round trip) -----------
// One buffer row per item: [delta of plane 0 | delta of plane 1 | ... | touched (1 / 0)], zero for rows this rank did not rewrite;
// the buffer is all-reduced (sum) as it stands, then every rank applies  cur = base + sum / count  (MEAN planes) or  base + sum
// (SUM planes) to the rows whose count is > 0, takes the result as the new base and clears its touched byte.
struct SyncPlanes { float* cur[12]; float* base[12]; int W[12]; int off[12]; int mean[12]; int n; int wsum; };
__global__ __launch_bounds__(256) void k_sync_dense_pack(SyncPlanes p, const unsigned char* touched, long long n_items, float* buf) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int ld = p.wsum + 1;
    if (e >= n_items * ld) return;
    const long long i = e / ld;
    const int c = (int)(e - i * ld);
    const bool t = touched[i] != 0;
    if (c == p.wsum) { buf[e] = t ? 1.f : 0.f; return; }
    int q = 0;
    while (q + 1 < p.n && c >= p.off[q + 1]) ++q;
    const size_t o = (size_t)i * p.W[q] + (c - p.off[q]);
    buf[e] = t ? p.cur[q][o] - p.base[q][o] : 0.f;
}
__global__ __launch_bounds__(256) void k_sync_dense_apply(SyncPlanes p, unsigned char* touched, long long n_items, const float* buf) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int ld = p.wsum + 1;
    if (e >= n_items * ld) return;
    const long long i = e / ld;
    const int c = (int)(e - i * ld);
    const float cnt = buf[i * ld + p.wsum];
    if (cnt <= 0.f) return;
    if (c == p.wsum) { touched[i] = 0; return; }
    int q = 0;
    while (q + 1 < p.n && c >= p.off[q + 1]) ++q;
    const size_t o = (size_t)i * p.W[q] + (c - p.off[q]);
    const float v = p.base[q][o] + (p.mean[q] && cnt > 1.f ? buf[e] / cnt : buf[e]);
    p.cur[q][o] = v;
    p.base[q][o] = v;
}

// ---- virtual ranks (g4r_virtual_train_steps): the dense-gradient buffers of n handles on one device summed in rank order -- what
// the RCCL all-reduce of a real n-GPU run delivers -- and handed back to every handle
struct VSumArgs { const float* src[16]; float* dst[16]; };
__global__ __launch_bounds__(256) void k_virtual_sum(VSumArgs a, int n, int count, float* tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int q = 0; q < n; ++q) s += a.src[q][i];      // rank order
    tmp[i] = s;
}
__global__ __launch_bounds__(256) void k_virtual_bcast(VSumArgs a, int n, int count, const float* tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float s = tmp[i];
    for (int q = 0; q < n; ++q) a.dst[q][i] = s;
}

// ---- one-shot all-reduce of the dense gradients through peer memory (g4r_p2p_*; the switch next to the RCCL all-reduce) --------
// The dense GRU gradients are a few hundred KB (cfg2: 60,600 floats): for that size a ring / tree collective is latency, not
// bandwidth -- every rank can simply READ the other ranks' gradients over its point-to-point xGMI links (7 x 240 KB in parallel
// at 8 ranks) and add them up itself, in rank order, so that every rank holds the same bits.  Every rank owns an exchange region
// (IPC-mapped by all peers): nblk stamps + two gradient buffers of `cap` floats (steps alternate between them).  Workgroup b of
// step s  (1) copies its 1024 floats of the local gradient to its own region, buffer s & 1, with system-scope write-through stores,
// drains them, and publishes stamp s + 1 in its own flag b;  (2) waits until every peer's flag b carries that stamp;  (3) loads the
// peers' 1024 floats (system-scope loads: no stale line of this GPU's L2 / L1) and sums own + peers in rank order into the local
// gradient buffer.  Hand-offs are per workgroup: nothing waits for a whole buffer, nothing waits for another workgroup of the same
// GPU (no residency assumption).  A buffer is rewritten two steps later, which needs every peer past step s + 1's wait, i.e. done
// reading step s.  The step number is the workgroup's own counter (round[b], local memory): replays of a captured graph need no
// host-side argument.  A peer that never arrives (a dead rank) ends the wait after `spin_ticks` of the 100 MHz wall clock and
// raises round[nblk]; the gradient is left as it was and g4r_train_steps reports the failure.
#define G4R_P2P_MAX 8
struct P2PArgs {
    float* data[G4R_P2P_MAX];          // [2][cap] of every rank as mapped in this process (own: the local pointer)
    unsigned* flags[G4R_P2P_MAX];      // [nblk]
    float* own_data;                   // = data[rank], flags[rank]: no dynamic index into the argument block
    unsigned* own_flags;
    unsigned* round;                   // local: [nblk] steps done by workgroup b, [nblk] = timeout marker
    int nranks, rank, count, cap, nblk;
    long long spin_ticks;
};
__device__ __forceinline__ void st4_sys(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }      // s_nop: guide 5.7 item 1 (stores)
__global__ __launch_bounds__(256) void k_p2p_allreduce(P2PArgs a, float* g) {
    __shared__ int bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.round[a.nblk]) return;                 // an earlier step gave up on a peer: the run is over, do not wait again
    const unsigned done = a.round[b];
    const unsigned stamp = done + 1;
    const size_t buf = (size_t)(done & 1) * a.cap;
    const int i = (b * 256 + tid) * 4;
    f32x4 mine = {0.f, 0.f, 0.f, 0.f};
    if (i + 3 < a.count) mine = *reinterpret_cast<const f32x4*>(g + i);
    else
        for (int k = 0; k < 4; ++k) if (i + k < a.count) mine[k] = g[i + k];
    if (tid == 0) bad = 0;
    st4_sys(a.own_data + buf + i, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope (the stores above are write-through already)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.own_flags + b, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < a.nranks && tid != a.rank) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(a.flags[tid] + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - stamp) < 0) {
            if (wall_clock64() - t0 > a.spin_ticks) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (bad) {
        if (tid == 0) a.round[a.nblk] = 1;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // every slot loads (slots past nranks and the own one read the own region); the eight loads and their wait are ONE asm statement
    // with early-clobber outputs (guide section 5.7 item 1, form (i)): hipcc takes an asm load's destination as written when the
    // statement ends and may copy it right away -- here that is true
    f32x4 v[G4R_P2P_MAX];
    const float* pp[G4R_P2P_MAX];
#pragma unroll
    for (int q = 0; q < G4R_P2P_MAX; ++q) pp[q] = (q < a.nranks ? a.data[q] : a.own_data) + buf + i;
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(pp[0]), "v"(pp[1]), "v"(pp[2]), "v"(pp[3]), "v"(pp[4]), "v"(pp[5]), "v"(pp[6]), "v"(pp[7]) : "memory");
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < G4R_P2P_MAX; ++q)
        if (q < a.nranks) s += (q == a.rank) ? mine : v[q];      // rank order, from 0.f like k_virtual_sum: the same bits on every rank
    if (i + 3 < a.count) *reinterpret_cast<f32x4*>(g + i) = s;
    else
        for (int k = 0; k < 4; ++k) if (i + k < a.count) g[i + k] = s[k];
    if (tid == 0) a.round[b] = stamp;
}
