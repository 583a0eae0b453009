// Probe (round 6): what one step's launch chain is made of at cfg #2 sizes, measured in the setting the step kernels run in -- a
// hipGraph of dependent launches on one stream, every kernel reading data that the launch in front of it rewrote from other XCDs.
//   k_empty            nothing                                             -> the launch floor
//   k_rt<N>            N dependent vector round trips (pointer chase through buffers the writer launch rewrote), one store
//   k_desc             descriptor pointer (kernarg) -> s_load pointers -> s_load state -> vector load: the step kernels' prologue
//   k_pull<KB>         every workgroup pulls the SAME KB of "weights" (rewritten by the writer) with dwordx4 loads, all in flight;
//                      rot = 1 starts every workgroup at a different offset (different L2 channels at the same time)
//   k_handoff          G groups of 7 workgroups: each publishes 1 KB as 8-byte {value, tag} granules (agent-scope write-through stores)
//                      and collects the 6 other tiles of its group (agent-scope loads, retried until every tag matches):
//                      the in-launch exchange a GRU forward split over 7 column workgroups per row block would need
// Durations: `rocprofv3 --kernel-trace --stats` on this binary (per kernel name) and the in-kernel s_memrealtime stamps it prints.
// build: hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }      // 100 MHz

__global__ __launch_bounds__(256) void k_writer(int* A, int* B, int* C, float* W, int nW, int gen) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 65536) { A[i] = (i * 7 + gen) & 65535; B[i] = (i * 13 + gen) & 65535; C[i] = (i * 29 + gen) & 65535; }
    for (int k = i; k < nW; k += gridDim.x * 256) W[k] = (float)((k + gen) & 1023) * 1e-3f;
}
__global__ __launch_bounds__(256) void k_empty(float* out) {
    if (out == nullptr) __builtin_trap();
}
template <int N>
__global__ __launch_bounds__(256) void k_rt(const int* A, const int* B, const int* C, int* out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    i = A[i & 65535];
    if (N >= 2) i = B[i];
    if (N >= 3) i = C[i];
    out[blockIdx.x * 256 + threadIdx.x] = i;
}
struct Desc { const int* A; const int* B; long long pad[14]; const int* state; int* out; };
__global__ __launch_bounds__(256) void k_desc(const Desc* __restrict__ d, int dummy) {
    const int* st = d->state;
    const int g = st[0];                                 // scalar: descriptor -> state
    const int* src = (g & 1) ? d->B : d->A;
    d->out[blockIdx.x * 256 + threadIdx.x] = src[(blockIdx.x * 256 + threadIdx.x + g) & 65535];
}
// N dependent scalar loads through read-only memory (each link in its own cache line): the price of a lazily read descriptor field
struct Link { const Link* next; long long pad[15]; };
template <int N>
__global__ __launch_bounds__(256) void k_schain(const Link* __restrict__ d, int* out) {
    const Link* p = d;
#pragma unroll
    for (int i = 0; i < N; ++i) p = p->next;
    out[blockIdx.x * 256 + threadIdx.x] = (int)(size_t)p;
}
// straight-line code of N x 64 VALU instructions (8 bytes each): what a launch pays for fetching cold instructions
template <int N>
__global__ __launch_bounds__(256) void k_code(float* out, float x) {
    float a = x + threadIdx.x, b = x;
#pragma unroll
    for (int i = 0; i < N * 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
template <int KB>
__global__ __launch_bounds__(512) void k_pull(const float* __restrict__ W, float* out, long long* trace, int rot) {
    const long long t0 = now();
    constexpr int NQ = KB * 64;            // float4 per workgroup
    constexpr int PER = (NQ + 511) / 512;
    float4 v[PER];
    const float4* w4 = (const float4*)W;
    const int start = rot ? ((blockIdx.x * 97) % NQ) : 0;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        int q = threadIdx.x + 512 * p;
        q = q < NQ ? q : NQ - 1;
        q += start; if (q >= NQ) q -= NQ;
        v[p] = w4[q];
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) s += v[p].x + v[p].y + v[p].z + v[p].w;
    __syncthreads();
    const long long t1 = now();
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0) { trace[2 * blockIdx.x] = t0; trace[2 * blockIdx.x + 1] = t1; }
}
// groups of 7: workgroup (grp = blockIdx.x % G, j = blockIdx.x / G) -- a group's members sit on one XCD when G is a multiple of 8
__global__ __launch_bounds__(512) void k_handoff(unsigned long long* xbuf, unsigned gen, int G, long long* trace, float* sink) {
    const int grp = blockIdx.x % G, j = blockIdx.x / G, tid = threadIdx.x;
    const long long t0 = now();
    unsigned long long* mine = xbuf + ((size_t)grp * 7 + j) * 256;
    if (tid < 256) {
        const float val = (float)(tid + j) * 0.5f;
        const unsigned long long gnl = ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(val);
        __hip_atomic_store(mine + tid, gnl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t1 = now();
    // collect: 6 x 256 granules over 512 threads = 3 per thread
    float acc = 0.f;
    int spins = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int e = tid + 512 * p;              // 0 .. 1535
        int src = e >> 8; if (src >= j) ++src;    // the other six
        const unsigned long long* gp = xbuf + ((size_t)grp * 7 + src) * 256 + (e & 255);
        unsigned long long gnl = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((unsigned)(gnl >> 32) != gen) {
            __builtin_amdgcn_s_sleep(1);
            gnl = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > (1 << 20)) break;
        }
        acc += __uint_as_float((unsigned)gnl);
    }
    __syncthreads();
    const long long t2 = now();
    if (acc == 1234.5f) sink[0] = acc;
    if (tid == 0) { trace[4 * blockIdx.x] = t0; trace[4 * blockIdx.x + 1] = t1; trace[4 * blockIdx.x + 2] = t2; trace[4 * blockIdx.x + 3] = spins; }
}

static void pct(const char* name, std::vector<double> v) {
    std::sort(v.begin(), v.end());
    printf("%-46s min %.2f  p50 %.2f  p90 %.2f  max %.2f us\n", name, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
}

int main() {
    int *A, *B, *C, *out; float *W, *fo; long long* trace; unsigned long long* xbuf; Desc* dd; int* state;
    const int nW = 64 * 1024;      // 256 KB of "weights"
    CK(hipMalloc(&A, 65536 * 4)); CK(hipMalloc(&B, 65536 * 4)); CK(hipMalloc(&C, 65536 * 4)); CK(hipMalloc(&out, 65536 * 4));
    CK(hipMalloc(&W, nW * 4)); CK(hipMalloc(&fo, 1024)); CK(hipMalloc(&trace, 4096 * 8)); CK(hipMalloc(&xbuf, 64 * 7 * 256 * 8));
    CK(hipMalloc(&dd, sizeof(Desc))); CK(hipMalloc(&state, 64));
    CK(hipMemset(xbuf, 0, 64 * 7 * 256 * 8)); CK(hipMemset(state, 0, 64));
    Desc h{}; h.A = A; h.B = B; h.state = state; h.out = out;
    CK(hipMemcpy(dd, &h, sizeof(Desc), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int REPS = 200;
    unsigned gen = 1;
    auto writer = [&]() { hipLaunchKernelGGL(k_writer, dim3(256), dim3(256), 0, s, A, B, C, W, nW, (int)gen); };
    auto timed = [&](const char* name, auto launch, int per_iter) {
        // graph of 16 x (writer, kernel): the same launch mechanism as the step
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 16; ++i) { writer(); for (int k = 0; k < per_iter; ++k) launch(); }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < REPS / 16; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-40s %7.2f us per (writer + %d x kernel)\n", name, ms * 1000 / ((REPS / 16) * 16), per_iter);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    timed("writer alone", [&]() {}, 0);
    timed("k_empty x1", [&]() { hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, fo); }, 1);
    timed("k_empty x5", [&]() { hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, fo); }, 5);
    timed("k_empty 256 wg x5", [&]() { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, fo); }, 5);
    timed("k_rt<1>", [&]() { hipLaunchKernelGGL(k_rt<1>, dim3(64), dim3(256), 0, s, A, B, C, out); }, 1);
    timed("k_rt<2>", [&]() { hipLaunchKernelGGL(k_rt<2>, dim3(64), dim3(256), 0, s, A, B, C, out); }, 1);
    timed("k_rt<3>", [&]() { hipLaunchKernelGGL(k_rt<3>, dim3(64), dim3(256), 0, s, A, B, C, out); }, 1);
    timed("k_desc", [&]() { hipLaunchKernelGGL(k_desc, dim3(64), dim3(256), 0, s, dd, 0); }, 1);
    {
        Link* links; CK(hipMalloc(&links, 64 * sizeof(Link)));
        std::vector<Link> hl(64);
        for (int i = 0; i < 64; ++i) hl[i].next = links + (i * 37 + 11) % 64;
        CK(hipMemcpy(links, hl.data(), 64 * sizeof(Link), hipMemcpyHostToDevice));
        timed("k_schain<1>", [&]() { hipLaunchKernelGGL(k_schain<1>, dim3(64), dim3(256), 0, s, links, out); }, 1);
        timed("k_schain<4>", [&]() { hipLaunchKernelGGL(k_schain<4>, dim3(64), dim3(256), 0, s, links, out); }, 1);
        timed("k_schain<8>", [&]() { hipLaunchKernelGGL(k_schain<8>, dim3(64), dim3(256), 0, s, links, out); }, 1);
        timed("k_schain<16>", [&]() { hipLaunchKernelGGL(k_schain<16>, dim3(64), dim3(256), 0, s, links, out); }, 1);
        timed("k_code<1> (64 instr)", [&]() { hipLaunchKernelGGL(k_code<1>, dim3(64), dim3(256), 0, s, fo, 1.0f); }, 1);
        timed("k_code<8> (512 instr, 4 KB)", [&]() { hipLaunchKernelGGL(k_code<8>, dim3(64), dim3(256), 0, s, fo, 1.0f); }, 1);
        timed("k_code<32> (2048 instr, 16 KB)", [&]() { hipLaunchKernelGGL(k_code<32>, dim3(64), dim3(256), 0, s, fo, 1.0f); }, 1);
    }
    auto pull_report = [&](const char* name, int nwg) {
        std::vector<long long> tr(2 * nwg);
        CK(hipMemcpy(tr.data(), trace, 16 * nwg, hipMemcpyDeviceToHost));
        std::vector<double> d; long long t0 = tr[0], t1 = tr[1];
        for (int i = 0; i < nwg; ++i) { d.push_back((tr[2 * i + 1] - tr[2 * i]) / 100.0); t0 = std::min(t0, tr[2 * i]); t1 = std::max(t1, tr[2 * i + 1]); }
        printf("   span %.2f us; ", (t1 - t0) / 100.0); pct(name, d);
    };
#define PULL(KB, NWG, ROT) { char nm[64]; snprintf(nm, 64, "k_pull<%d KB> x %d wg rot %d", KB, NWG, ROT); \
        timed(nm, [&]() { hipLaunchKernelGGL(k_pull<KB>, dim3(NWG), dim3(512), 0, s, W, fo, trace, ROT); }, 1); pull_report(nm, NWG); }
    PULL(32, 56, 0) PULL(32, 56, 1) PULL(32, 128, 0) PULL(32, 128, 1)
    PULL(48, 56, 0) PULL(48, 56, 1) PULL(48, 128, 1)
    PULL(96, 32, 0) PULL(96, 32, 1)
    PULL(144, 32, 0) PULL(144, 32, 1)
    PULL(240, 8, 0)
    for (int G : {8, 16}) {
        // eager launches with a fresh tag each (a graph would freeze the tag)
        std::vector<double> pub, col, tot; long long worst_spin = 0;
        for (int it = 0; it < 50; ++it) {
            ++gen; writer();
            hipLaunchKernelGGL(k_handoff, dim3(7 * G), dim3(512), 0, s, xbuf, gen, G, trace, fo);
            CK(hipStreamSynchronize(s));
            std::vector<long long> tr(4 * 7 * G);
            CK(hipMemcpy(tr.data(), trace, 32 * 7 * G, hipMemcpyDeviceToHost));
            if (it < 5) continue;
            long long tmin = tr[0];
            for (int i = 0; i < 7 * G; ++i) tmin = std::min(tmin, tr[4 * i]);
            for (int i = 0; i < 7 * G; ++i) {
                pub.push_back((tr[4 * i + 1] - tr[4 * i]) / 100.0); col.push_back((tr[4 * i + 2] - tr[4 * i + 1]) / 100.0);
                tot.push_back((tr[4 * i + 2] - tmin) / 100.0); worst_spin = std::max(worst_spin, tr[4 * i + 3]);
            }
        }
        printf("k_handoff, %d groups of 7 (worst spin count %lld)\n", G, worst_spin);
        pct("   publish issue", pub); pct("   collect (publish issued -> all six tiles)", col); pct("   first workgroup start -> this one done", tot);
    }
    return 0;
}
