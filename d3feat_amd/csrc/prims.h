// Fused building blocks of the preprocessing pipelines (grid subsampling, neighbour grid).
//
// A fragment is ~400 launches, three quarters of them tiny preprocessing kernels; inside a replayed HIP graph every
// node costs a few microseconds of dispatch whatever its size.  These templates cut the node count:
//   * begin_kernel      one launch for all "reset" work of an op: up to four fills, lens -> offsets, bounding-box
//                       initialisation, ticket counters;
//   * bbox_kernel<Epi>  per-element bounding boxes + an epilogue run by the LAST workgroup to finish (ticket counter), e.g.
//                       the per-element grid geometry that needs the complete boxes;
//   * scan_fold_kernel<In, Epi>  exclusive scan in ONE launch: every workgroup scans its 1024-item tile (`local`), the last
//                       workgroup to finish scans the tile sums into `base` and runs an epilogue with the grand total.
//                       Consumers read  local[i] + base[i / D3F_SCAN_TILE]  (d3f_scan_at) instead of a third "add" pass.
// Inter-workgroup hand-off ("last workgroup") follows cdna_hip_programming.md G16: every workgroup's stores, then
// __syncthreads, then one lane: agent-scope release fence + s_waitcnt + device-scope atomic ticket; the last arriver
// issues an agent-scope acquire fence before its (first-touch) reads of the other workgroups' results.
#pragma once
#include "common.h"

#define D3F_SCAN_TILE 1024

struct D3fFill {
    unsigned* p;
    unsigned long long n;   // 32-bit words
    unsigned v;
};

__device__ __forceinline__ bool d3f_last_block(unsigned* counter, unsigned nblocks) {
    __shared__ int last_flag;
    // every wave waits for ITS OWN stores to be acknowledged before the barrier: the workgroup-scope release inside
    // __syncthreads may omit vmcnt(0) outside tgsplit mode, and lane 0's agent-scope release below only covers its own wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = atomicAdd(counter, 1u);
        last_flag = (t == nblocks - 1u) ? 1 : 0;
        if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return last_flag != 0;
}

// offs[0..B] = prefix sums of lens; bbox[b*6 + {0,1,2}] = 0xFFFFFFFF (min slots), {3,4,5} = 0 (max slots); counters = 0;
// fills f0..f3 (n = 0: unused).  Any grid size; 256 threads.
static __global__ void __launch_bounds__(256) begin_kernel(const int* __restrict__ lens, int B, int* __restrict__ offs,
                                                    unsigned* __restrict__ bbox, unsigned* __restrict__ counters,
                                                    int ncounters, D3fFill f0, D3fFill f1, D3fFill f2, D3fFill f3) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0 && lens) {
            int s = 0;
            for (int b = 0; b < B; ++b) { offs[b] = s; s += lens[b]; }
            offs[B] = s;
        }
        if (bbox)
            for (int i = threadIdx.x; i < B * 6; i += blockDim.x) bbox[i] = ((i % 6) < 3) ? 0xFFFFFFFFu : 0u;
        for (int i = threadIdx.x; i < ncounters; i += blockDim.x) counters[i] = 0u;
    }
    const D3fFill f[4] = {f0, f1, f2, f3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned* p = f[k].p;
        const size_t n = (size_t)f[k].n;
        if (n == 0) continue;
        const size_t n4 = (((uintptr_t)p & 15) == 0) ? n / 4 : 0;
        const uint4 v4 = make_uint4(f[k].v, f[k].v, f[k].v, f[k].v);
        for (size_t j = tid; j < n4; j += stride) ((uint4*)p)[j] = v4;
        for (size_t j = n4 * 4 + tid; j < n; j += stride) p[j] = f[k].v;
    }
}

static inline int d3f_begin_launch(const int* lens, int B, int* offs, unsigned* bbox, unsigned* counters, int ncounters,
                                   D3fFill f0, D3fFill f1, D3fFill f2, D3fFill f3, hipStream_t stream) {
    const unsigned long long words = f0.n + f1.n + f2.n + f3.n;
    long long blocks = (long long)(words / 4 / 256) + 1;
    if (blocks > 1024) blocks = 1024;
    begin_kernel<<<(int)blocks, 256, 0, stream>>>(lens, B, offs, bbox, counters, ncounters, f0, f1, f2, f3);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// Bounding boxes per batch element, grid (chunks, B); min / max are exact and order independent, so the result equals
// cpp_utils/cloud/cloud.cpp:27-66 bit for bit.  epi(b_first, b_step) is run by the last workgroup with all 256 threads.
// ptrs != NULL: element b's points are read IN PLACE from ptrs[b] (its own array, rows 0 .. len_b - 1) instead of from rows
// [offs[b], offs[b + 1]) of the stacked array `pts` (the stage-0 clouds of a replay are never copied into one buffer).
template <class Epi>
__global__ void __launch_bounds__(256) bbox_kernel(const float* __restrict__ pts, const float* const* __restrict__ ptrs,
                                                   const int* __restrict__ offs, int B,
                                                   unsigned* __restrict__ bbox, unsigned* __restrict__ counter, Epi epi) {
    const int b = blockIdx.y;
    const int lo = offs[b], hi = offs[b + 1];
    if (ptrs) pts = ptrs[b] - 3 * (size_t)lo;          // virtual base: row i of the stack = row i - lo of the element
    unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    // four points per thread and trip: twelve loads in flight (one point per trip was a chain of dependent HBM round trips --
    // 60 us for the 1.2 M points of a stage-0 stack)
    const int stride = gridDim.x * blockDim.x;
    for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += 4 * stride) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int iu = min(i + u * stride, hi - 1);       // (a repeated point changes no minimum)
#pragma unroll
            for (int d = 0; d < 3; ++d) v[u][d] = pts[3 * (size_t)iu + d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned x = d3f_f2ord(v[u][d]);
                mn[d] = min(mn[d], x);
                mx[d] = max(mx[d], x);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[d] = min(mn[d], (unsigned)__shfl_xor((int)mn[d], o, 64));
            mx[d] = max(mx[d], (unsigned)__shfl_xor((int)mx[d], o, 64));
        }
    }
    // one atomic per block and slot (same-address atomics serialise in L2 at ~12 ns each)
    __shared__ unsigned red[4][6];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[wv][d] = mn[d]; red[wv][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6 && lo < hi) {
        unsigned v = red[0][threadIdx.x];
        for (int k = 1; k < 4; ++k) v = (threadIdx.x < 3) ? min(v, red[k][threadIdx.x]) : max(v, red[k][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[b * 6 + threadIdx.x], v);
        else atomicMax(&bbox[b * 6 + threadIdx.x], v);
    }
    if (d3f_last_block(counter, gridDim.x * gridDim.y)) epi();
}

template <class Epi>
static inline int d3f_bbox_launch_t(const float* pts, const int* offs, int B, int N, unsigned* bbox, unsigned* counter, Epi epi,
                                    hipStream_t stream, const float* const* ptrs = nullptr) {
    // about one workgroup per CU over all elements: every workgroup ends with a ticket on ONE counter, and same-address
    // atomics serialise at ~12 ns each, so thousands of (mostly idle) workgroups cost more than the boxes themselves
    // (N is the whole stack: an element holds about N / B points; eight points per thread)
    int chunks = d3f_cdiv(d3f_cdiv(N > 0 ? N : 1, B), 256 * 8);
    const int per_elem = 512 / B > 1 ? 512 / B : 1;
    if (chunks > per_elem) chunks = per_elem;
    bbox_kernel<Epi><<<dim3(chunks, B), 256, 0, stream>>>(pts, ptrs, offs, B, bbox, counter, epi);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ---- one-launch exclusive scan --------------------------------------------------------------------------------------
__device__ __forceinline__ int d3f_block_excl_scan_256(int v, int* lds4, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) lds4[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = lds4[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__device__ __forceinline__ int d3f_scan_at(const int* __restrict__ local, const int* __restrict__ base, int i) {
    return local[i] + base[i / D3F_SCAN_TILE];
}

// in(i) -> item i (i < n).  local[i] = exclusive scan inside i's tile; base[t] = sum of the tiles before t (written by the
// last workgroup); epi(total) runs in the last workgroup (all 256 threads) after `base` is complete.
template <class In, class Epi>
__global__ void __launch_bounds__(256) scan_fold_kernel(In in, int n, const int* __restrict__ n_dev, int* __restrict__ local,
                                                        int* __restrict__ base, unsigned* __restrict__ counter, Epi epi) {
    __shared__ int lds[4];
    n = d3f_dyn(n, n_dev);   // items beyond the device-side count are zeros that are neither read nor written
    // capacity-sized launch: only the tiles that hold items take part (and take a ticket); tile 0 always does, so that the
    // epilogue runs for an empty input too
    const int live = max(1, (n + D3F_SCAN_TILE - 1) / D3F_SCAN_TILE);
    if ((int)blockIdx.x >= live) return;
    const int i0 = blockIdx.x * D3F_SCAN_TILE + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (i0 + k < n) ? in(i0 + k) : 0;
        s += v[k];
    }
    int tot;
    int ex = d3f_block_excl_scan_256(s, lds, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) local[i0 + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) base[blockIdx.x] = tot;
    if (!d3f_last_block(counter, (unsigned)live)) return;
    const int nblocks = live;
    int carry = 0;
    for (int c = 0; c < nblocks; c += 256) {
        const int i = c + threadIdx.x;
        const int x = (i < nblocks) ? __hip_atomic_load(&base[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int t;
        const int e = d3f_block_excl_scan_256(x, lds, &t);
        if (i < nblocks) base[i] = carry + e;
        carry += t;
    }
    __syncthreads();
    epi(carry);
}

struct D3fScanIn {   // plain array input
    const int* p;
    __device__ __forceinline__ int operator()(int i) const { return p[i]; }
};
struct D3fNoEpi {
    __device__ __forceinline__ void operator()(int) const {}
    __device__ __forceinline__ void operator()() const {}
};

static inline size_t d3f_scan_base_ints(int n) { return (size_t)d3f_cdiv(n > 0 ? n : 1, D3F_SCAN_TILE) + 64; }

template <class In, class Epi>
static inline int d3f_scan_fold_launch(In in, int n, const int* n_dev, int* local, int* base, unsigned* counter, Epi epi,
                                       hipStream_t stream) {
    const int nb = d3f_cdiv(n > 0 ? n : 1, D3F_SCAN_TILE);
    scan_fold_kernel<In, Epi><<<nb, 256, 0, stream>>>(in, n, n_dev, local, base, counter, epi);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
