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
