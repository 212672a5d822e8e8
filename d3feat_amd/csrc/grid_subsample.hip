// Voxel-grid barycentre subsampling on gfx950, bit-identical to the reference INCLUDING output row order.
//
// Reference: tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:5-97 (one cloud) and :101-149
// (batch); cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-105 (features / classes).
//
// The reference walks the points once, keeps one accumulator per voxel in a std::unordered_map and emits
// the map in iteration order.  Three things must be reproduced exactly:
//   (1) voxel keys: fp32 origin = floor(min*(1/dl))*dl, per-axis index floor((p-origin)/dl) with a true
//       IEEE divide, key = ix + NX*iy + NX*NY*iz in 64 bits;
//   (2) barycentre = fp32 sum of the voxel's points IN INPUT ORDER times (float)(1.0/count);
//   (3) row order = libstdc++ unordered_map iteration order for keys inserted in first-occurrence order.
//
// MI355X design (all HBM/L2-bound integer + gather work, no MFMA):
//   * keys -> open-addressing hash table in HBM (atomicCAS), first-occurrence index by atomicMin;
//   * first-occurrence flags -> exclusive scan = voxel ids in insertion order;
//   * per-voxel point lists in input order WITHOUT a sort: lock-free chains (atomicExch), then each point
//     ranks itself inside its chain and scatters its index -> a thread per voxel sums sequentially;
//   * (3) has a closed form: inserting a sequence into a fresh table of nb buckets yields the list
//     "buckets by DEscending first-insertion position, inside a bucket by DEscending position", and a
//     rehash re-inserts the current list order.  So the order is ~log2(M) rounds of
//     {bucket first-position (atomicMin), bucket sizes, reverse scan, rank inside bucket chain}, each fully
//     parallel: rounds up to 1109 buckets run in ONE 1024-thread workgroup per batch element, larger rounds grid-wide
//     (insert / tile scan + tile sums in its last workgroup / place);
//   * 9 + 3*rounds launches per call (csrc/prims.h: fused reset, boxes + geometry, one-launch scans); in capacity mode
//     (d3f_batch_grid_subsample_async) all sizes stay on the device and an overflowing call reports an empty result.
#include <cstdlib>
#include "prims.h"
#include "radix_sort.h"

#define GS_EMPTY 0xFFFFFFFFFFFFFFFFull
#define GS_KEYBITS 56
#define GS_KEYMASK ((1ull << GS_KEYBITS) - 1ull)
#define GS_MAX_PROBE 4096

__constant__ unsigned long long D3F_CHAIN_DEV[D3F_NCHAIN] = {
    13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull, 20753ull, 42043ull, 85229ull,
    172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull,
    101473717ull, 206062531ull, 418451333ull, 849749479ull, 1725587117ull, 3504151727ull};

struct GsElem {
    float org[3];
    int pad0;
    unsigned long long NX, NY, NZ;
    long long bbase;  // base of this element's bucket scratch
};

static unsigned long long gs_chain_ge(long long n) {
    for (int j = 0; j < D3F_NCHAIN; ++j)
        if ((long long)D3F_CHAIN_HOST[j] >= n) return D3F_CHAIN_HOST[j];
    return D3F_CHAIN_HOST[D3F_NCHAIN - 1];
}

__device__ __forceinline__ unsigned long long gs_mix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// ---- per-element origin / grid dims (grid_subsampling.cpp:24-30), one thread per element -------------
// Also the description of the stage-0 sort (smeta != NULL): the sort key is (element << kb) | voxel key, kb = bits of the
// largest cell count of any element, so only kb + bits(B - 1) key bits are significant -- 21 for four 3DMatch rooms (3 digit
// passes), 27 for a KITTI pair (4).  A key wider than 32 bits is reported (D3F_ST_KEY_WIDTH) with an empty result; the
// caller's synchronous path (hash form, any key < 2^56) takes it.
__device__ __forceinline__ int gs_bits(unsigned long long x) {   // bits needed to hold values 0..x
    int b = 0;
    while (x) { ++b; x >>= 1; }
    return b;
}
__device__ __forceinline__ void gs_prep(const unsigned* __restrict__ bbox, const int* __restrict__ offs, int B, float dl,
                                        GsElem* __restrict__ el, int* __restrict__ status, RsMeta* __restrict__ smeta,
                                        int n_cap) {
    __shared__ unsigned long long sCells[256];
    const int b = threadIdx.x;   // run by ONE 256-thread workgroup (B <= 255)
    unsigned long long cells = 0ull;
    if (b < B) {
        GsElem e;
        e.pad0 = 0;
        const int len = offs[b + 1] - offs[b];
        if (len <= 0) {
            atomicOr(&status[1], D3F_ST_EMPTY_ELEMENT);
            e.org[0] = e.org[1] = e.org[2] = 0.f;
            e.NX = e.NY = e.NZ = 1;
        } else {
            const float inv = __fdiv_rn(1.0f, dl);  // `1/sampleDl`
            float mx[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float mn = d3f_ord2f(__hip_atomic_load(&bbox[b * 6 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                mx[d] = d3f_ord2f(__hip_atomic_load(&bbox[b * 6 + 3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                e.org[d] = __fmul_rn(floorf(__fmul_rn(mn, inv)), dl);
            }
            e.NX = (unsigned long long)fmaxf(floorf(__fdiv_rn(__fsub_rn(mx[0], e.org[0]), dl)), 0.f) + 1ull;
            e.NY = (unsigned long long)fmaxf(floorf(__fdiv_rn(__fsub_rn(mx[1], e.org[1]), dl)), 0.f) + 1ull;
            e.NZ = (unsigned long long)fmaxf(floorf(__fdiv_rn(__fsub_rn(mx[2], e.org[2]), dl)), 0.f) + 1ull;
        }
        // bucket scratch base: prefix over elements of chain_ge(len)
        long long base = 0;
        for (int j = 0; j < b; ++j) {
            long long l = offs[j + 1] - offs[j];
            unsigned long long nb = D3F_CHAIN_DEV[D3F_NCHAIN - 1];
            for (int c = 0; c < D3F_NCHAIN; ++c)
                if ((long long)D3F_CHAIN_DEV[c] >= l) { nb = D3F_CHAIN_DEV[c]; break; }
            base += (long long)nb;
        }
        e.bbase = base;
        el[b] = e;
        // cells of the element's grid, saturated (each factor < 2^32 only when the product is meaningful as a sort key)
        const double c = (double)e.NX * (double)e.NY * (double)e.NZ;
        cells = c >= 1.8e19 ? 0xFFFFFFFFFFFFFFFFull : e.NX * e.NY * e.NZ;
    }
    if (!smeta) return;
    sCells[threadIdx.x] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long mx = 1ull;
        for (int j = 0; j < B; ++j) mx = sCells[j] > mx ? sCells[j] : mx;
        int kb = gs_bits(mx - 1ull);
        if (kb < 1) kb = 1;
        const int eb = gs_bits((unsigned long long)(B - 1));
        RsMeta m;
        m.n = min(offs[B], n_cap);
        m.kb = kb;
        m.bits = kb + eb;
        if (m.bits > 32) {      // does not fit the 32-bit sort key: nothing is sorted, the result is reported empty
            atomicOr(&status[1], D3F_ST_KEY_WIDTH);
            m.n = 0;
            m.bits = 8;
        }
        m.npass = (m.bits + 7) / 8;
        *smeta = m;
    }
}
// epilogue of the bounding-box kernel (last workgroup): origin / grid dims of every element from the finished boxes
struct GsPrepEpi {
    const unsigned* bbox; const int* offs; int B; float dl; GsElem* el; int* status; RsMeta* smeta; int n_cap;
    __device__ __forceinline__ void operator()() const { gs_prep(bbox, offs, B, dl, el, status, smeta, n_cap); }
};

// ---- voxel key per point + hash insert (grid_subsampling.cpp:49-59) ----------------------------------
__global__ void __launch_bounds__(256) gs_insert_kernel(const float* __restrict__ pts, int N, const int* __restrict__ offs,
                                                        int B, float dl, const GsElem* __restrict__ el,
                                                        unsigned long long* __restrict__ tkey, unsigned* __restrict__ tfirst,
                                                        unsigned long long capmask, int* __restrict__ slot,
                                                        int* __restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(N, offs[B])) return;   // N is the capacity, offs[B] the real point count
    const int b = d3f_find_elem(offs, B, i);
    const GsElem e = el[b];
    const float fx = floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + 0], e.org[0]), dl));
    const float fy = floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + 1], e.org[1]), dl));
    const float fz = floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + 2], e.org[2]), dl));
    int st = 0;
    if (fx < 0.f || fy < 0.f || fz < 0.f) st |= D3F_ST_NEG_CELL;
    const unsigned long long ix = (unsigned long long)fmaxf(fx, 0.f), iy = (unsigned long long)fmaxf(fy, 0.f),
                             iz = (unsigned long long)fmaxf(fz, 0.f);
    const unsigned long long key = ix + e.NX * iy + e.NX * e.NY * iz;
    if (key > GS_KEYMASK) st |= D3F_ST_KEY_RANGE;
    if (st) atomicOr(&status[1], st);
    const unsigned long long word = ((unsigned long long)b << GS_KEYBITS) | (key & GS_KEYMASK);
    unsigned long long h = gs_mix(word) & capmask;
    // linear probing, bounded: the table is sized for the caller's voxel capacity; more voxels than that fill it up and are
    // reported (D3F_ST_OUT_OVERFLOW -> empty result, recomputed by the caller) instead of probing forever
    bool placed = false;
    for (int probe = 0; probe < GS_MAX_PROBE; ++probe) {
        unsigned long long prev = atomicCAS(&tkey[h], GS_EMPTY, word);
        if (prev == GS_EMPTY || prev == word) { placed = true; break; }
        h = (h + 1) & capmask;
    }
    if (!placed) atomicOr(&status[1], D3F_ST_OUT_OVERFLOW);
    atomicMin(&tfirst[h], (unsigned)i);
    slot[i] = (int)h;
}

// first-occurrence flag of point i, the input of the voxel-id scan (fused into the scan kernel)
struct GsMarkIn {
    const int* n_dev; const int* slot; const unsigned* tfirst;
    __device__ __forceinline__ int operator()(int i) const { return (i < *n_dev && tfirst[slot[i]] == (unsigned)i) ? 1 : 0; }
};

// ---- voxel ids, voxel keys, per-voxel chains ----------------------------------------------------------
__global__ void __launch_bounds__(256) gs_chain_kernel(int N, const int* __restrict__ slot, const unsigned* __restrict__ tfirst,
                                                       const unsigned long long* __restrict__ tkey,
                                                       const int* __restrict__ vscan, const int* __restrict__ vbase,
                                                       int* __restrict__ pvid,
                                                       unsigned long long* __restrict__ vkey, int* __restrict__ vhead,
                                                       int* __restrict__ vcnt, int* __restrict__ pnext,
                                                       const int* __restrict__ n_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(N, *n_dev)) return;
    const int s = slot[i];
    const int fi = (int)tfirst[s];
    const int v = d3f_scan_at(vscan, vbase, fi);
    pvid[i] = v;
    if (fi == i) vkey[v] = tkey[s] & GS_KEYMASK;
    pnext[i] = atomicExch(&vhead[v], i);
    atomicAdd(&vcnt[v], 1);
}

// number of voxels created by the points before point p: hash form = the first-occurrence scan itself; sort form = the
// first-occurrence flags are BITS (one word per 32 points), scanned by popcount
struct GsRankScan {
    const int* vscan; const int* vbase;
    __device__ __forceinline__ int operator()(int p) const { return d3f_scan_at(vscan, vbase, p); }
};
struct GsRankBits {
    const unsigned* fbits; const int* wscan; const int* wbase;
    __device__ __forceinline__ int operator()(int p) const {
        const int w = p >> 5;
        return d3f_scan_at(wscan, wbase, w) + __popc(fbits[w] & ((1u << (p & 31)) - 1u));
    }
};
// epilogue of the voxel-id scan (last workgroup, total = M): per-element voxel offsets, the reported lengths and status
template <class Rank>
struct GsMoffsEpi {
    const int* offs; int B; Rank rank; int* meta; int* moffs; int* sub_lens; int* status_dev;
    int out_cap;
    int elem_cap;   // capacity mode: no element may hold more voxels than this (the order rounds are launched for it)
    __device__ __forceinline__ void operator()(int M) const {
        if (threadIdx.x != 0) return;   // B <= 255: one thread
        meta[0] = M;
        const int N = offs[B];
        // scan value at offs[b] = number of voxels created by points before element b (empty tail -> M)
        for (int b = 0; b <= B; ++b) moffs[b] = (b == B || offs[b] >= N) ? M : rank(offs[b]);
        // More voxels than the caller's output rows (capacity mode): flag it and report an EMPTY result, so that every
        // downstream stage of a captured launch sequence runs on zero rows instead of on partially written ones.
        bool over = M > out_cap ||
                    (__hip_atomic_load(&meta[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (D3F_ST_OUT_OVERFLOW | D3F_ST_KEY_WIDTH));
        for (int b = 0; b < B; ++b) over = over || (moffs[b + 1] - moffs[b] > elem_cap);
        if (over && !(__hip_atomic_load(&meta[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & D3F_ST_KEY_WIDTH))
            atomicOr(&meta[1], D3F_ST_OUT_OVERFLOW);
        for (int b = 0; b < B; ++b) {
            const int l = over ? 0 : moffs[b + 1] - moffs[b];
            meta[2 + b] = l;
            sub_lens[b] = l;
        }
        if (status_dev) {
            status_dev[0] = over ? 0 : M;   // rows valid in sub_points (none when it overflowed)
            status_dev[1] = __hip_atomic_load(&meta[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

// ---- rank of each point inside its voxel chain (= number of chain members with a smaller index) ----------
__global__ void __launch_bounds__(256) gs_rank_kernel(int N, const int* __restrict__ n_dev, const int* __restrict__ pvid,
                                                      const int* __restrict__ vhead, const int* __restrict__ pnext,
                                                      const int* __restrict__ vstart, const int* __restrict__ sbase,
                                                      int* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(N, *n_dev)) return;
    const int v = pvid[i];
    int r = 0;
    for (int j = vhead[v]; j >= 0; j = pnext[j]) r += (j < i) ? 1 : 0;
    sorted[d3f_scan_at(vstart, sbase, v) + r] = i;
}

// ---- libstdc++ unordered_map iteration order (closed form, see file header) --------------------------
__device__ __forceinline__ int gs_ld(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// key % nb for key < 2^56, nb < 2^31 without the 64-bit software division: double-precision quotient estimate
// (off by at most 1 or 2) and a branch-free correction.
__device__ __forceinline__ int gs_mod(unsigned long long key, unsigned nb, double inv_nb) {
    const unsigned long long q = (unsigned long long)((double)key * inv_nb);
    long long r = (long long)(key - q * (unsigned long long)nb);
    r += (r < 0) ? (long long)nb : 0;
    r += (r < 0) ? (long long)nb : 0;
    r -= (r >= (long long)nb) ? (long long)nb : 0;
    r -= (r >= (long long)nb) ? (long long)nb : 0;
    return (int)r;
}

// Round j reads the insertion sequence from list buffer P[j&1] and writes the new list order to P[(j+1)&1]; its
// bucket arrays are the parity-(j&1) set.  Rounds 0..GS_SMALL_LAST (bucket counts 13..1109) are tiny and run inside
// ONE single-workgroup launch (gs_order_small_kernel); a single CU however sustains only ~1 divergent L2 access per
// clock, so the large rounds (the last 2-5 for 10^4..10^5 voxels) are spread over the whole chip as four small
// grid-wide launches each (insert / tile scan / tile-sum scan / place).
#define GS_SMALL_LAST 6
#define GS_TILE 1024

struct GsOrderArgs {
    const unsigned long long* vkey;
    const int* moffs;
    const int* offs;
    const GsElem* el;
    int* P[2];      // list buffers, element b at + offs[b]
    int* nx[2];     // chain links, per position (parity sets: place(j) reads round j's while inserting round j + 1's)
    int* cd;        // reverse-scan values, per position
    int* bkt[2];    // bucket of the element at a position (parity sets)
    int* bf[2];     // per bucket: first insertion position   (parity sets, element b at + el[b].bbase)
    int* bc[2];     // per bucket: size
    int* bh[2];     // per bucket: chain head (last insertion)
    int* tsum;      // tile sums of the reverse scan, element b at + offs[b]/GS_TILE + b
    int* vpos;      // OUT: final position of each voxel inside its element
    int max_m;      // largest element the grid-wide rounds were launched for (capacity mode); larger ones are skipped
};

// All state of the small rounds (<= 1109 buckets, <= 1109 list entries) lives in LDS: the rounds are a chain of
// {reset, insert with atomics, reverse scan, chain-rank placement} phases separated by barriers, and with the bucket arrays in
// HBM every phase paid L2 round trips (225 us for the seven rounds of a 30 k-voxel cloud, on every fragment's critical path);
// with LDS atomics a phase costs a few hundred cycles.  Only the final list (and the finished positions) go to memory.
#define GS_SMALL_NB 1109   // D3F_CHAIN[GS_SMALL_LAST]
// 256 threads, not 1024: a workgroup of 16 wavefronts + 44 KB of LDS has to wait for a CU that the other replays in flight have
// left completely free (rocprof showed 190 us for 15 us of work); four wavefronts are placed at once
#define GS_SMALL_T 256
__global__ void __launch_bounds__(GS_SMALL_T) gs_order_small_kernel(GsOrderArgs A, int small_last) {
    __shared__ int wsum[GS_SMALL_T / 64];
    __shared__ int sBF[GS_SMALL_NB], sBC[GS_SMALL_NB], sBH[GS_SMALL_NB], sNX[GS_SMALL_NB], sCD[GS_SMALL_NB], sBK[GS_SMALL_NB];
    __shared__ int sL[2][GS_SMALL_NB];
    __shared__ unsigned long long sKey[GS_SMALL_NB];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int M = A.moffs[b + 1] - A.moffs[b];
    if (M <= 0) return;
    const unsigned long long* key = A.vkey + A.moffs[b];
    int* VP = A.vpos + A.moffs[b];
    const long long bbase = A.el[b].bbase;
    if (small_last > GS_SMALL_LAST) small_last = GS_SMALL_LAST;     // the LDS arrays are sized for 1109 buckets
    for (int t = tid; t < min(M, GS_SMALL_NB); t += GS_SMALL_T) sKey[t] = key[t];
    int lo = 0, cur = 0;
    bool done = false;
    __syncthreads();
    for (int j = 0; j <= small_last && j < D3F_NCHAIN; ++j) {
        const int nb = (int)D3F_CHAIN_DEV[j];
        const double inv_nb = 1.0 / (double)nb;
        const bool last = M <= nb;
        const int hi = last ? M : nb;
        const int* La = sL[cur];
        int* Lb = sL[cur ^ 1];
        for (int t = tid; t < nb; t += GS_SMALL_T) { sBF[t] = 0x7fffffff; sBC[t] = 0; sBH[t] = -1; }
        __syncthreads();
        for (int t = tid; t < hi; t += GS_SMALL_T) {
            const int id = (t < lo) ? La[t] : t;
            const int bk = gs_mod(sKey[id], (unsigned)nb, inv_nb);
            atomicMin(&sBF[bk], t);
            atomicAdd(&sBC[bk], 1);
            sNX[t] = atomicExch(&sBH[bk], t);
            sBK[t] = bk;
        }
        __syncthreads();
        int carry = 0;
        for (int c0 = 0; c0 < hi; c0 += GS_SMALL_T) {
            const int u = c0 + tid, t = hi - 1 - u;
            int c = 0;
            if (u < hi) {
                const int bk = sBK[t];
                c = (sBF[bk] == t) ? sBC[bk] : 0;
            }
            int x = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                int y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (lane == 63) wsum[w] = x;
            __syncthreads();
            int wbase = 0, tot = 0;
#pragma unroll
            for (int q = 0; q < GS_SMALL_T / 64; ++q) {
                int v = wsum[q];
                if (q < w) wbase += v;
                tot += v;
            }
            __syncthreads();
            if (u < hi) sCD[t] = carry + wbase + x - c;
            carry += tot;
        }
        __syncthreads();
        for (int t = tid; t < hi; t += GS_SMALL_T) {
            const int id = (t < lo) ? La[t] : t;
            const int bk = sBK[t];
            int r = 0;
            for (int q = sBH[bk]; q >= 0; q = sNX[q]) r += (q > t) ? 1 : 0;
            const int dest = sCD[sBF[bk]] + r;
            Lb[dest] = id;
            if (last) VP[id] = dest;
        }
        __syncthreads();
        cur ^= 1;
        if (last) { done = true; break; }
        lo = hi;
    }
    if (!done && small_last + 1 < D3F_NCHAIN) {
        // hand over to the first grid-wide round j: it reads the current list from P[j & 1] and expects clean bucket arrays
        const int j = small_last + 1;
        int* P = A.P[j & 1] + A.offs[b];
        for (int t = tid; t < lo; t += GS_SMALL_T) P[t] = sL[cur][t];
        const int nb = (int)D3F_CHAIN_DEV[j];
        int* BF = A.bf[j & 1] + bbase;
        int* BC = A.bc[j & 1] + bbase;
        int* BH = A.bh[j & 1] + bbase;
        for (int t = tid; t < nb; t += GS_SMALL_T) { BF[t] = 0x7fffffff; BC[t] = 0; BH[t] = -1; }
    }
}

// geometry of grid-wide round j for element b; false when the element finished in an earlier round
__device__ __forceinline__ bool gs_round(const GsOrderArgs& A, int b, int j, int& M, int& lo, int& hi, int& nb, bool& last) {
    M = A.moffs[b + 1] - A.moffs[b];
    lo = (int)D3F_CHAIN_DEV[j - 1];
    if (M <= lo || M > A.max_m) return false;   // (an element beyond the launch geometry is flagged D3F_ST_OUT_OVERFLOW)
    nb = (int)D3F_CHAIN_DEV[j];
    last = M <= nb;
    hi = last ? M : nb;
    return true;
}

__global__ void __launch_bounds__(256) gs_order_insert_kernel(GsOrderArgs A, int j) {
    const int b = blockIdx.y;
    int M, lo, hi, nb;
    bool last;
    if (!gs_round(A, b, j, M, lo, hi, nb, last)) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= hi) return;
    const int o = A.offs[b];
    const long long bbase = A.el[b].bbase;
    const int id = (t < lo) ? A.P[j & 1][o + t] : t;
    const int bk = gs_mod(A.vkey[A.moffs[b] + id], (unsigned)nb, 1.0 / (double)nb);
    atomicMin(&A.bf[j & 1][bbase + bk], t);
    atomicAdd(&A.bc[j & 1][bbase + bk], 1);
    A.nx[j & 1][o + t] = atomicExch(&A.bh[j & 1][bbase + bk], t);
    A.bkt[j & 1][o + t] = bk;
}

// exclusive scan of one element's tile sums (run by ONE workgroup of 256 threads)
__device__ __forceinline__ void gs_order_scan_sums(const GsOrderArgs& A, int j, int b, int* wsum) {
    int M, lo, hi, nb;
    bool last;
    if (!gs_round(A, b, j, M, lo, hi, nb, last)) return;
    int* ts = A.tsum + A.offs[b] / GS_TILE + b;
    const int ntiles = (hi + GS_TILE - 1) / GS_TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int carry = 0;
    for (int c0 = 0; c0 < ntiles; c0 += 256) {
        const int i = c0 + tid;
        const int v = (i < ntiles) ? __hip_atomic_load(&ts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int s = wsum[q];
            if (q < w) wbase += s;
            tot += s;
        }
        __syncthreads();
        if (i < ntiles) ts[i] = carry + wbase + x - v;
        carry += tot;
    }
}

// tile-local reverse exclusive scan of c[t] = (t first of its bucket) ? bucket size : 0, over u = hi-1-t; the last
// workgroup of EACH ELEMENT then scans that element's tile sums (gs_order_place adds them).  Tickets are per element and
// only the tiles that hold items take one: the elements finish independently instead of queueing behind one workgroup,
// and the idle tiles of a capacity-sized launch do not serialise on the counter.
__global__ void __launch_bounds__(256) gs_order_scan_tiles_kernel(GsOrderArgs A, int j, int B, unsigned* __restrict__ counter) {
    __shared__ int wsum[4];
    const int b = blockIdx.y;
    int M, lo, hi, nb;
    bool last;
    const int tile = blockIdx.x;
    const bool active = gs_round(A, b, j, M, lo, hi, nb, last) && tile * GS_TILE < hi;   // workgroup-uniform
    if (!active) return;
    {
        const int o = A.offs[b];
        const long long bbase = A.el[b].bbase;
        const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
        if (!last && j + 1 < D3F_NCHAIN) {
            // clear the other parity's bucket arrays: round j + 1 is inserted by this round's place kernel (nb_next < 2.24 hi;
            // they were last read by round j - 1's place kernel, which has finished)
            const int nbn = (int)D3F_CHAIN_DEV[j + 1];
            const int stride = ((hi + GS_TILE - 1) / GS_TILE) * 256;
            for (int i = tile * 256 + tid; i < nbn; i += stride) {
                A.bf[(j + 1) & 1][bbase + i] = 0x7fffffff;
                A.bc[(j + 1) & 1][bbase + i] = 0;
                A.bh[(j + 1) & 1][bbase + i] = -1;
            }
        }
        int c[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = tile * GS_TILE + tid * 4 + k, t = hi - 1 - u;
            c[k] = 0;
            if (u < hi) {
                const int bk = A.bkt[j & 1][o + t];
                c[k] = (A.bf[j & 1][bbase + bk] == t) ? A.bc[j & 1][bbase + bk] : 0;
            }
            s += c[k];
        }
        int x = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int v = wsum[q];
            if (q < w) wbase += v;
            tot += v;
        }
        int run = wbase + x - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = tile * GS_TILE + tid * 4 + k;
            if (u < hi) A.cd[o + hi - 1 - u] = run;
            run += c[k];
        }
        if (tid == 0) A.tsum[o / GS_TILE + b + tile] = tot;
    }
    if (!d3f_last_block(counter + b, (unsigned)((hi + GS_TILE - 1) / GS_TILE))) return;
    gs_order_scan_sums(A, j, b, wsum);
}

// Placement of round j -- and, for an element that goes on, the INSERTION of round j + 1 in the same launch: the element placed
// at `dest` is position `dest` of the next round's sequence, the new voxels nb_j .. hi' - 1 follow at their own positions
// (two launches per grid-wide round instead of three).
__global__ void __launch_bounds__(256) gs_order_place_kernel(GsOrderArgs A, int j) {
    const int b = blockIdx.y;
    int M, lo, hi, nb;
    bool last;
    if (!gs_round(A, b, j, M, lo, hi, nb, last)) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int o = A.offs[b];
    const long long bbase = A.el[b].bbase;
    const bool next = !last && j + 1 < D3F_NCHAIN;
    const int nbn = next ? (int)D3F_CHAIN_DEV[j + 1] : 1;
    int id = -1, pos = -1;          // what this thread inserts into round j + 1: voxel `id` at position `pos`
    if (t < hi) {
        id = (t < lo) ? A.P[j & 1][o + t] : t;
        const int bk = A.bkt[j & 1][o + t];
        int r = 0;
        for (int q = A.bh[j & 1][bbase + bk]; q >= 0; q = A.nx[j & 1][o + q]) r += (q > t) ? 1 : 0;
        const int f = A.bf[j & 1][bbase + bk];
        const int dest = A.cd[o + f] + A.tsum[o / GS_TILE + b + (hi - 1 - f) / GS_TILE] + r;
        A.P[(j + 1) & 1][o + dest] = id;
        if (last) A.vpos[A.moffs[b] + id] = dest;
        pos = dest;
    } else if (next && t < min(M, nbn)) {
        id = pos = t;               // (hi == nb_j here: the voxels created after the rehash)
    }
    if (next && id >= 0) {
        const int bk = gs_mod(A.vkey[A.moffs[b] + id], (unsigned)nbn, 1.0 / (double)nbn);
        atomicMin(&A.bf[(j + 1) & 1][bbase + bk], pos);
        atomicAdd(&A.bc[(j + 1) & 1][bbase + bk], 1);
        A.nx[(j + 1) & 1][o + pos] = atomicExch(&A.bh[(j + 1) & 1][bbase + bk], pos);
        A.bkt[(j + 1) & 1][o + pos] = bk;
    }
}

#include "gs_small.h"

// ---- sort form of the point -> voxel pass (capacity mode, large clouds: the stage-0 call) ---------------------------------
// A STABLE radix sort of (element, voxel key) -> the points of a voxel become one run, in input order (what the in-order
// barycentre needs), its first entry is the voxel's first occurrence.  Replaces hash insert / first-occurrence gather-scan /
// chains / count scan / in-chain rank / accumulation -- six passes of random atomics and gathers over every raw point -- by:
//   gs_sortkey_kernel   keys from the coordinates + digit-0 histogram of the tile            (reads 12 B, writes 4 B per point)
//   rs_* (radix_sort.h) 3 digit passes for a 21-bit key (2 launches each, 16 B per point and pass)
//   gs_heads_kernel     run heads of the sorted keys -> first-occurrence BITS (one atomicOr per voxel)
//   scan_fold           popcount scan of the bit words = voxel ids in insertion order; offsets / lengths / status in its epilogue
//   gs_runs_kernel      the head of a run walks it: in-order fp32 sum, barycentre, voxel key -> per-voxel records
//   (iteration-order rounds on the voxel keys, as in the hash form)
//   gs_emit_kernel      barycentres to their rows
// Results are identical to the hash form (same voxel ids, keys, per-voxel point order).
__global__ void __launch_bounds__(RS_THREADS) gs_sortkey_kernel(const float* __restrict__ pts, const float* const* __restrict__ ptrs,
                                                                const int* __restrict__ offs, int B,
                                                                float dl, const GsElem* __restrict__ el,
                                                                const RsMeta* __restrict__ smeta, unsigned* __restrict__ skey,
                                                                unsigned* __restrict__ hist, int* __restrict__ status) {
    __shared__ unsigned sHist[256];
    __shared__ int sOffs[D3F_MAX_BATCH + 1];
    __shared__ GsElem sEl[D3F_MAX_BATCH];
    __shared__ const float* sBase[D3F_MAX_BATCH];     // virtual base of every element: row i of the stack = sBase[b] + 3 i
    const int n = smeta->n, kb = smeta->kb;
    const int tile = blockIdx.x;
    if ((long long)tile * RS_TILE >= (long long)n) return;
    // the element table goes through LDS: a per-point search through `offs` in memory was a chain of dependent loads
    for (int t = threadIdx.x; t <= B; t += RS_THREADS) sOffs[t] = offs[t];
    for (int t = threadIdx.x; t < B; t += RS_THREADS) {
        sEl[t] = el[t];
        sBase[t] = ptrs ? ptrs[t] - 3 * (size_t)offs[t] : pts;       // (in place: the clouds of a replay are never stacked)
    }
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = tile * RS_TILE + w * RS_WAVE_ITEMS + lane;
    // the wave's 1024 consecutive points usually belong to one cloud: one search for the wave, a per-point search otherwise
    const int wlo = tile * RS_TILE + w * RS_WAVE_ITEMS, whi = min(wlo + RS_WAVE_ITEMS, n) - 1;
    int b0 = 0;
    while (b0 + 1 < B && wlo >= sOffs[b0 + 1]) ++b0;
    const bool one_cloud = (b0 + 1 >= B) || whi < sOffs[b0 + 1];
    unsigned key[RS_ROUNDS], vm = 0u;
    int st = 0;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int i = base + r * 64;
        key[r] = 0u;
        if (i < n) {
            int b = b0;
            if (!one_cloud)
                while (b + 1 < B && i >= sOffs[b + 1]) ++b;
            const float ox = sEl[b].org[0], oy = sEl[b].org[1], oz = sEl[b].org[2];
            // 32-bit index arithmetic: this kernel only runs when NX NY NZ <= 2^kb <= 2^32 (gs_prep; otherwise n == 0), and
            // the bounding box keeps every index below its dimension
            const unsigned NX = (unsigned)sEl[b].NX, NY = (unsigned)sEl[b].NY;
            const float* __restrict__ P = sBase[b];
            const float fx = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 0], ox), dl));
            const float fy = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 1], oy), dl));
            const float fz = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 2], oz), dl));
            if (fx < 0.f || fy < 0.f || fz < 0.f) st |= D3F_ST_NEG_CELL;
            const unsigned ix = (unsigned)fmaxf(fx, 0.f), iy = (unsigned)fmaxf(fy, 0.f), iz = (unsigned)fmaxf(fz, 0.f);
            const unsigned k = ix + NX * (iy + NY * iz);                 // < NX NY NZ <= 2^kb
            key[r] = (kb < 32 ? ((unsigned)b << kb) : 0u) | (kb < 32 ? (k & ((1u << kb) - 1u)) : k);
            skey[i] = key[r];
            vm |= 1u << r;
        }
    }
    if (st) atomicOr(&status[1], st);
    rs_tile_histogram(key, vm, 0, sHist, hist + (size_t)tile * 256);
}

// run heads of the sorted keys: bit i of fbits <- point i is the first point of its voxel (the sort is stable)
__global__ void __launch_bounds__(256) gs_heads_kernel(int N, const RsMeta* __restrict__ smeta, const unsigned* __restrict__ key0,
                                                       const unsigned* __restrict__ key1, const unsigned* __restrict__ val0,
                                                       const unsigned* __restrict__ val1, unsigned* __restrict__ fbits) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= min(N, smeta->n)) return;
    const unsigned* __restrict__ ks = (smeta->npass & 1) ? key1 : key0;
    const unsigned* __restrict__ vs = (smeta->npass & 1) ? val1 : val0;
    const unsigned k = ks[j];
    if (j == 0 || ks[j - 1] != k) {
        const unsigned i = vs[j];
        atomicOr(&fbits[i >> 5], 1u << (i & 31u));
    }
}
// input of the voxel-id scan: first occurrences per word of 32 points
struct GsBitsIn {
    const unsigned* fbits;
    __device__ __forceinline__ int operator()(int w) const { return __popc(fbits[w]); }
};
// one thread per sorted position; the head of a run owns the voxel: id by bit rank, key, in-order barycentre
// (grid_subsampling.cpp:63-70, :81-92: fp32 sum in input order x (float)(1.0 / count)).
// Every thread first fetches ITS position's point (index -> coordinates: independent loads, one round trip for the whole
// workgroup) and parks it in LDS; the head then sums its run out of LDS -- a run that crosses the workgroup's last position goes
// on through memory (about one head in 25).  Before, every head walked its ~10 points as a chain of dependent
// index -> coordinate loads: 56 us for the 1.2 M points of a stage-0 stack.
__global__ void __launch_bounds__(256) gs_runs_kernel(int N, const RsMeta* __restrict__ smeta, const unsigned* __restrict__ key0,
                                                      const unsigned* __restrict__ key1, const unsigned* __restrict__ val0,
                                                      const unsigned* __restrict__ val1, GsRankBits rank,
                                                      const float* __restrict__ pts, const float* const* __restrict__ ptrs,
                                                      const int* __restrict__ offs, unsigned long long* __restrict__ vkey,
                                                      float* __restrict__ bary) {
    __shared__ float sx[256], sy[256], sz[256];
    __shared__ unsigned sk[256];
    const int n = min(N, smeta->n);
    const int j0 = blockIdx.x * 256;
    if (j0 >= n) return;
    const int t = threadIdx.x, j = j0 + t;
    const unsigned* __restrict__ ks = (smeta->npass & 1) ? key1 : key0;
    const unsigned* __restrict__ vs = (smeta->npass & 1) ? val1 : val0;
    unsigned k = 0u, pi = 0u, kprev = 0u;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (j < n) {
        k = ks[j];
        pi = vs[j];
        kprev = j > 0 ? ks[j - 1] : 0u;
        if (ptrs) {              // in place: the element is the key's upper field, its array starts at stack row offs[b]
            const int b = smeta->kb < 32 ? (int)(k >> smeta->kb) : 0;
            pts = ptrs[b] - 3 * (size_t)offs[b];
        }
        px = pts[3 * (size_t)pi + 0]; py = pts[3 * (size_t)pi + 1]; pz = pts[3 * (size_t)pi + 2];
    }
    sx[t] = px; sy[t] = py; sz[t] = pz;
    sk[t] = k;
    __syncthreads();
    if (j >= n || (j != 0 && kprev == k)) return;
    const int v = rank((int)pi);
    float ax = px, ay = py, az = pz;
    int c = 1;
    const int lim = min(256, n - j0);                   // positions of this workgroup that exist
    while (t + c < lim && sk[t + c] == k) {
        ax = __fadd_rn(ax, sx[t + c]);
        ay = __fadd_rn(ay, sy[t + c]);
        az = __fadd_rn(az, sz[t + c]);
        ++c;
    }
    if (t + c == 256) {                                 // the run may go on in the next workgroup's positions (same key: same element)
        while (j + c < n && ks[j + c] == k) {
            const size_t q = vs[j + c];
            ax = __fadd_rn(ax, pts[3 * q + 0]);
            ay = __fadd_rn(ay, pts[3 * q + 1]);
            az = __fadd_rn(az, pts[3 * q + 2]);
            ++c;
        }
    }
    const float sc = (float)(1.0 / (double)c);  // `1.0 / v.second.count` is a double, narrowed by operator*(PointXYZ, float)
    vkey[v] = (unsigned long long)(smeta->kb < 32 ? (k & ((1u << smeta->kb) - 1u)) : k);
    bary[3 * (size_t)v + 0] = __fmul_rn(ax, sc);
    bary[3 * (size_t)v + 1] = __fmul_rn(ay, sc);
    bary[3 * (size_t)v + 2] = __fmul_rn(az, sc);
}
// barycentre of voxel v -> output row (element offset + iteration-order position)
__global__ void __launch_bounds__(256) gs_emit_kernel(int* __restrict__ status, const int* __restrict__ moffs, int B,
                                                      const int* __restrict__ vpos, const float* __restrict__ bary,
                                                      float* __restrict__ out_p, int out_cap) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= status[0]) return;
    const int b = d3f_find_elem(moffs, B, v);
    const size_t dest = (size_t)moffs[b] + (size_t)vpos[v];
    if (dest >= (size_t)out_cap) {   // (cannot happen once the epilogue has checked M: kept as the last line of defence)
        atomicOr(&status[1], D3F_ST_OUT_OVERFLOW);
        return;
    }
    out_p[3 * dest + 0] = bary[3 * (size_t)v + 0];
    out_p[3 * dest + 1] = bary[3 * (size_t)v + 1];
    out_p[3 * dest + 2] = bary[3 * (size_t)v + 2];
}


// ---- per-voxel in-order accumulation + emit (grid_subsampling.cpp:63-70, :81-92) -----------------------
__global__ void __launch_bounds__(256) gs_accum_kernel(const float* __restrict__ pts, const float* __restrict__ feat,
                                                       int fdim, int* __restrict__ status,
                                                       const int* __restrict__ moffs, int B,
                                                       const int* __restrict__ vstart, const int* __restrict__ sbase,
                                                       const int* __restrict__ vcnt,
                                                       const int* __restrict__ sorted, const int* __restrict__ vpos,
                                                       float* __restrict__ out_p, float* __restrict__ out_f, int out_cap) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= status[0]) return;
    const int b = d3f_find_elem(moffs, B, v);
    // sbase == NULL: vstart holds the run starts themselves (sort form); otherwise the tiled scan of the counts
    const int n = vcnt[v], st = sbase ? d3f_scan_at(vstart, sbase, v) : vstart[v];
    const size_t dest = (size_t)moffs[b] + (size_t)vpos[v];
    if (dest >= (size_t)out_cap) {   // more voxels than the caller's output rows (capacity mode): report, never write
        atomicOr(&status[1], D3F_ST_OUT_OVERFLOW);
        return;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int t = 0; t < n; ++t) {
        const size_t i = (size_t)sorted[st + t];
        sx = __fadd_rn(sx, pts[3 * i + 0]);
        sy = __fadd_rn(sy, pts[3 * i + 1]);
        sz = __fadd_rn(sz, pts[3 * i + 2]);
    }
    const float sc = (float)(1.0 / (double)n);  // `1.0 / v.second.count` is a double, narrowed by operator*(PointXYZ, float)
    out_p[3 * dest + 0] = __fmul_rn(sx, sc);
    out_p[3 * dest + 1] = __fmul_rn(sy, sc);
    out_p[3 * dest + 2] = __fmul_rn(sz, sc);
    if (fdim > 0) {
        const float cf = (float)n;
        for (int f = 0; f < fdim; ++f) {
            float s = 0.f;
            for (int t = 0; t < n; ++t) s = __fadd_rn(s, feat[(size_t)sorted[st + t] * fdim + f]);
            out_f[dest * fdim + f] = __fdiv_rn(s, cf);
        }
    }
}

// classes: the reference's max_element over unordered_map<int,int> compares (label, count) pairs, label
// first, i.e. returns the LARGEST label id present in the voxel (grid_subsampling.cpp:94).
__global__ void __launch_bounds__(256) gs_fill_kernel(int* __restrict__ p, size_t n, int v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) gs_labels_kernel(int N, int ldim, const int* __restrict__ cls,
                                                        const int* __restrict__ pvid, const int* __restrict__ moffs, int B,
                                                        const int* __restrict__ vpos, int* __restrict__ out_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int v = pvid[i];
    const int b = d3f_find_elem(moffs, B, v);
    const size_t dest = (size_t)moffs[b] + (size_t)vpos[v];
    for (int l = 0; l < ldim; ++l) atomicMax(&out_c[dest * ldim + l], cls[(size_t)i * ldim + l]);
}

// ------------------------------------------------------------------------------------------------
struct GsLayout {
    size_t cap;
    long long bucket_total;
};
// keys: the hash table holds one entry per VOXEL, so it is sized by the voxel capacity where the caller states one (capacity
// mode: M_cap << N for the stage-0 call -- 155 k voxels of 1.26 M raw points -- a 6 MB table that stays in L2 instead of a
// 50 MB one, for the three passes of random atomics / gathers that go through it); load factor <= 0.5 either way.
static GsLayout gs_layout(int N, int B, int keys = -1) {
    GsLayout L;
    L.cap = 64;
    const size_t nk = (size_t)((keys > 0 && keys < N) ? keys : (N > 0 ? N : 1));
    while (L.cap < (size_t)2 * nk) L.cap <<= 1;
    // sum_b chain_ge(len_b) <= chain_ge-ratio bound: every chain step is < 2.24x, so chain_ge(l) < 2.24*l + 13
    L.bucket_total = (long long)(2.24 * (double)N) + 16ll * B + 64;
    return L;
}

extern "C" size_t d3f_grid_subsample_workspace_bytes(int N, int B, int fdim, int ldim) {
    (void)fdim; (void)ldim;
    if (N < 0 || B < 1) return 0;
    GsLayout L = gs_layout(N, B);
    size_t n = (size_t)(N > 0 ? N : 1);
    size_t bytes = 0;
    bytes += d3f_align((B + 1) * sizeof(int)) * 2 + d3f_align((B + 2) * sizeof(int));   // offs, moffs, meta
    bytes += d3f_align(B * 6 * sizeof(unsigned));           // bbox
    bytes += d3f_align(B * sizeof(GsElem));
    bytes += d3f_align(L.cap * sizeof(unsigned long long)); // tkey
    bytes += d3f_align(L.cap * sizeof(int));                // tfirst
    bytes += d3f_align(n * sizeof(unsigned long long));     // vkey
    bytes += 16 * d3f_align(n * sizeof(int));               // slot vscan pvid vhead vcnt pnext vstart sorted vpos L0 L1 nx(2) cd bkt(2)
    bytes += 6 * d3f_align((size_t)L.bucket_total * sizeof(int));
    bytes += d3f_align((n / GS_TILE + B + 8) * sizeof(int));
    bytes += 2 * d3f_align(d3f_scan_base_ints(N) * sizeof(int)) + d3f_align((4 + D3F_NCHAIN * (size_t)B) * sizeof(unsigned));
    // sort form: digit histograms, first-occurrence bits, per-voxel barycentres, the sort's description
    bytes += d3f_align(rs_hist_words(N) * sizeof(unsigned)) + d3f_align((n / 32 + 2) * sizeof(unsigned)) +
             d3f_align(3 * n * sizeof(float)) + d3f_align(sizeof(RsMeta));
    // one-workgroup form: staging blocks [B][min(N, 16384)][3] + per-cloud counts / flags
    bytes += d3f_align(7 * (size_t)B * (n < 16384 ? n : 16384) * sizeof(float)) + 256 + 2 * d3f_align(B * sizeof(int));
    return bytes + 4096;
}

// One-workgroup-per-cloud form (gs_small.h): every cloud of the stack has at most `pc` <= 16384 points.
template <int T, int R>
static int gs_small_launch(const GsSmallArgs& A, float* sub_points, int* sub_lens_dev, int* status_dev, hipStream_t stream) {
    static std::atomic<unsigned long long> lds_done{0ull};
    const void* const fns[] = {(const void*)gs_small_kernel<T, R>};
    if (d3f_opt_in_lds(lds_done, fns, (int)gss_lds_bytes<T, R>(GSS_NB_MAX)) != D3F_OK) return D3F_ERR_HIP;
    gs_small_kernel<T, R><<<A.B, T, gss_lds_bytes<T, R>(A.nbmax), stream>>>(A);
    D3F_LAUNCH_CHECK();
    gs_small_pack_kernel<<<d3f_cdiv(A.out_cap, 256), 256, 0, stream>>>(A, sub_points, sub_lens_dev, status_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
static int gs_run_small(const float* points, int N, const int* lens_dev, int B, float dl, float* sub_points, int M_cap, int elem_cap,
                        int pc, int* sub_lens_dev, int* status_dev, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (elem_cap <= 0 || elem_cap > M_cap) elem_cap = M_cap;
    if (elem_cap > pc) elem_cap = pc;                   // voxels <= points
    D3fArena ar(workspace, workspace_bytes);
    GsSmallArgs A;
    A.stage = ar.take<float>(3 * (size_t)B * (size_t)elem_cap);
    A.recs = ar.take<float>(4 * (size_t)B * (size_t)elem_cap);
    A.mcount = ar.take<int>(B);
    A.cflags = ar.take<int>(B);
    if (!ar.ok) return D3F_ERR_WORKSPACE;
    A.pts = points; A.lens = lens_dev; A.B = B; A.dl = dl; A.elem_cap = elem_cap; A.out_cap = M_cap;
    // the largest iteration-order round that has to fit the workgroup's LDS: the first chain value >= the voxel capacity
    A.nbmax = GSS_NB_MAX;
    for (int j = 0; j < D3F_NCHAIN; ++j)
        if ((long long)D3F_CHAIN_HOST[j] >= (long long)elem_cap) { A.nbmax = (int)D3F_CHAIN_HOST[j]; break; }
    if (A.nbmax > GSS_NB_MAX) A.nbmax = GSS_NB_MAX;      // (a cloud with more voxels is reported: D3F_ST_OUT_OVERFLOW)
    if (A.nbmax < 1109) A.nbmax = 1109;
    if (pc <= 2048) return gs_small_launch<256, 8>(A, sub_points, sub_lens_dev, status_dev, stream);
    if (pc <= 4096) return gs_small_launch<512, 8>(A, sub_points, sub_lens_dev, status_dev, stream);
    if (pc <= 8192) return gs_small_launch<1024, 8>(A, sub_points, sub_lens_dev, status_dev, stream);
    if (pc <= 12288) return gs_small_launch<1024, 12>(A, sub_points, sub_lens_dev, status_dev, stream);
    return gs_small_launch<1024, 16>(A, sub_points, sub_lens_dev, status_dev, stream);
}

// Shared implementation.  sync mode (status_host != NULL): ONE host synchronisation after the voxel count is known,
// everything after it sized by M.  async mode (status_dev != NULL): no host round trip at all -- N and M_cap are
// capacities, the real sizes live in HBM (lens_dev -> offs[B]; M -> status_dev[0]) and every kernel bounds itself by them,
// so the call can be captured in a HIP graph and replayed for clouds of any size up to the capacity.
static int gs_run(const float* points, int N, const int* lens_dev, int B, float dl, const float* features, int fdim,
                  const int* classes, int ldim, float* sub_points, int M_cap, int elem_cap, int elem_points, float* sub_features,
                  int* sub_classes, int* sub_lens_dev, int* status_host, int* status_dev, void* workspace,
                  size_t workspace_bytes, hipStream_t stream, const float* const* ptrs = nullptr) {
    const bool async = status_dev != nullptr;
    if (async && !features && ldim == 0 && !ptrs) {
        // one workgroup per cloud when the caller's capacities fit it: every cloud <= 16384 points and <= 5087 voxels (a cloud
        // beyond its stated capacity is reported by either form, so the choice adds no failure mode)
        const int pc = (elem_points > 0 && elem_points < N) ? elem_points : N;
        int ec = (elem_cap > 0 && elem_cap < M_cap) ? elem_cap : M_cap;
        if (ec > pc) ec = pc;
        if (pc <= 16384 && ec <= GSS_NB_MAX) return gs_run_small(points, N, lens_dev, B, dl, sub_points, M_cap, elem_cap, pc, sub_lens_dev, status_dev,
                                             workspace, workspace_bytes, stream);
    }
    GsLayout L = gs_layout(N, B, async ? M_cap : -1);
    D3fArena ar(workspace, workspace_bytes);
    const size_t n = (size_t)N;
    int* offs = ar.take<int>(B + 1);
    int* moffs = ar.take<int>(B + 1);
    unsigned* bbox = ar.take<unsigned>(B * 6);
    GsElem* el = ar.take<GsElem>(B);
    // ticket counters: [0] boxes, [1] voxel-id scan, [2] start scan, [4 + j * B + b] iteration-order round j of element b
    const int ncounters = 4 + D3F_NCHAIN * B;
    unsigned* counters = ar.take<unsigned>(ncounters);
    // [tkey | tfirst | vhead] are reset to 0xFFFFFFFF and [meta | vcnt] to 0 by ONE launch: keep each group contiguous
    unsigned long long* tkey = ar.take<unsigned long long>(L.cap);
    unsigned* tfirst = ar.take<unsigned>(L.cap);
    int* vhead = ar.take<int>(n);
    int* meta = ar.take<int>(B + 2);   // [M, flags, sub_lens...]
    int* vcnt = ar.take<int>(n);
    unsigned long long* vkey = ar.take<unsigned long long>(n);
    int* slot = ar.take<int>(n);
    int* vscan = ar.take<int>(n);
    int* pvid = ar.take<int>(n);
    int* pnext = ar.take<int>(n);
    int* vstart = ar.take<int>(n);
    int* sorted = ar.take<int>(n);
    GsOrderArgs A;
    A.vpos = ar.take<int>(n);
    A.P[0] = ar.take<int>(n);
    A.P[1] = ar.take<int>(n);
    A.nx[0] = ar.take<int>(n);
    A.nx[1] = ar.take<int>(n);
    A.cd = ar.take<int>(n);
    A.bkt[0] = ar.take<int>(n);
    A.bkt[1] = ar.take<int>(n);
    for (int k = 0; k < 2; ++k) {
        A.bf[k] = ar.take<int>((size_t)L.bucket_total);
        A.bc[k] = ar.take<int>((size_t)L.bucket_total);
        A.bh[k] = ar.take<int>((size_t)L.bucket_total);
    }
    A.tsum = ar.take<int>(n / GS_TILE + B + 8);
    int* vbase = ar.take<int>(d3f_scan_base_ints(N));   // tile offsets of the voxel-id scan
    int* sbase = ar.take<int>(d3f_scan_base_ints(N));   // ... of the voxel-start scan
    // sort form of the point -> voxel pass: capacity mode, points only, large calls (see gs_sortkey_kernel)
    const bool use_sort = async && !features && ldim == 0;     // capacity mode: always the sort form (or gs_run_small above)
    unsigned* rs_hist = ar.take<unsigned>(rs_hist_words(N));
    unsigned* fbits = ar.take<unsigned>(n / 32 + 2);
    float* bary = ar.take<float>(3 * n);
    RsMeta* smeta = ar.take<RsMeta>(1);
    if (!ar.ok) return D3F_ERR_WORKSPACE;
    A.vkey = vkey; A.moffs = moffs; A.offs = offs; A.el = el;
    if (elem_cap <= 0 || elem_cap > M_cap) elem_cap = M_cap;
    if (elem_cap > N) elem_cap = N;
    A.max_m = async ? elem_cap : 0x7fffffff;

    // 9 + 3 * rounds launches (was 27 + 4 * rounds): reset -> boxes (+ grid geometry) -> keys / hash insert ->
    // voxel-id scan (first-occurrence flags fused in, per-element offsets / lengths / status in its last workgroup) ->
    // chains -> start scan -> in-chain rank -> iteration order (small rounds in one workgroup, large rounds grid-wide) ->
    // in-order accumulation.
    int rc;
    const D3fFill none{nullptr, 0ull, 0u};
    const unsigned long long ones_words = (unsigned long long)((char*)(vhead + n) - (char*)tkey) / 4ull;
    const unsigned long long zero_words = (unsigned long long)((char*)(vcnt + n) - (char*)meta) / 4ull;
    // (sort form: only the first-occurrence bits and the status words need clearing)
    if ((rc = use_sort ? d3f_begin_launch(lens_dev, B, offs, bbox, counters, ncounters, D3fFill{fbits, (unsigned long long)(n / 32 + 2), 0u},
                                          D3fFill{(unsigned*)meta, (unsigned long long)(B + 2), 0u}, none, none, stream)
                       : d3f_begin_launch(lens_dev, B, offs, bbox, counters, ncounters, D3fFill{(unsigned*)tkey, ones_words, 0xFFFFFFFFu},
                                          D3fFill{(unsigned*)meta, zero_words, 0u}, none, none, stream)) != D3F_OK) return rc;
    GsPrepEpi prep{bbox, offs, B, dl, el, meta, use_sort ? smeta : nullptr, N};
    if ((rc = d3f_bbox_launch_t(points, offs, B, N, bbox, counters, prep, stream, ptrs)) != D3F_OK) return rc;
    const int nblk = d3f_cdiv(N, 256);
    const int elem_lim = async ? elem_cap : 0x7fffffff;
    int M, maxM;       // sizes of the voxel-indexed launches
    if (use_sort) {
        // storage reuse: sort keys in slot / pnext, point indices in pvid / sorted, word scan in vscan / vbase
        unsigned *key0 = (unsigned*)slot, *key1 = (unsigned*)pnext, *val0 = (unsigned*)pvid, *val1 = (unsigned*)sorted;
        gs_sortkey_kernel<<<rs_tiles(N), RS_THREADS, 0, stream>>>(points, ptrs, offs, B, dl, el, smeta, key0, rs_hist, meta);
        D3F_LAUNCH_CHECK();
        if ((rc = rs_sort_launch(smeta, N, key0, key1, val0, val1, rs_hist, stream)) != D3F_OK) return rc;
        gs_heads_kernel<<<nblk, 256, 0, stream>>>(N, smeta, key0, key1, val0, val1, fbits);
        D3F_LAUNCH_CHECK();
        const GsRankBits rank{fbits, vscan, vbase};
        GsMoffsEpi<GsRankBits> mepi{offs, B, rank, meta, moffs, sub_lens_dev, status_dev, M_cap, elem_lim};
        if ((rc = d3f_scan_fold_launch(GsBitsIn{fbits}, N / 32 + 1, nullptr, vscan, vbase, counters + 1, mepi, stream)) != D3F_OK)
            return rc;
        gs_runs_kernel<<<nblk, 256, 0, stream>>>(N, smeta, key0, key1, val0, val1, rank, points, ptrs, offs, vkey, bary);
        D3F_LAUNCH_CHECK();
        M = N < M_cap ? N : M_cap;
        maxM = elem_cap;
    } else {
    GsMoffsEpi<GsRankScan> mepi{offs, B, GsRankScan{vscan, vbase}, meta, moffs, sub_lens_dev, status_dev, M_cap, elem_lim};
    gs_insert_kernel<<<nblk, 256, 0, stream>>>(points, N, offs, B, dl, el, tkey, tfirst, (unsigned long long)L.cap - 1ull,
                                               slot, meta);
    D3F_LAUNCH_CHECK();
    if ((rc = d3f_scan_fold_launch(GsMarkIn{offs + B, slot, tfirst}, N, offs + B, vscan, vbase, counters + 1, mepi, stream)) != D3F_OK)
        return rc;
    if (!async) {
        // The output size is data dependent (as for the reference op, whose output tensor is allocated after the
        // computation): ONE host synchronisation here brings back M, the flags and the per-element counts; everything
        // after it is sized by M instead of N.
        D3F_HIP_TRY(hipMemcpyAsync(status_host, meta, (B + 2) * sizeof(int), hipMemcpyDeviceToHost, stream));
        D3F_HIP_TRY(hipStreamSynchronize(stream));
        M = status_host[0];
        if (status_host[1] != 0 || M <= 0) return D3F_OK;  // flags are reported to the caller; nothing more to compute
        maxM = 0;
        for (int b = 0; b < B; ++b) maxM = status_host[2 + b] > maxM ? status_host[2 + b] : maxM;
    } else {
        M = N < M_cap ? N : M_cap;   // upper bound: voxels <= points, and the caller promises <= M_cap (checked on device)
        maxM = elem_cap;             // ... and <= elem_cap in any one element: the order rounds are launched for that
    }

    gs_chain_kernel<<<nblk, 256, 0, stream>>>(N, slot, tfirst, tkey, vscan, vbase, pvid, vkey, vhead, vcnt, pnext, offs + B);
    D3F_LAUNCH_CHECK();
    if ((rc = d3f_scan_fold_launch(D3fScanIn{vcnt}, async ? N : M, meta, vstart, sbase, counters + 2, D3fNoEpi{}, stream)) != D3F_OK)
        return rc;
    gs_rank_kernel<<<nblk, 256, 0, stream>>>(N, offs + B, pvid, vhead, pnext, vstart, sbase, sorted);
    }
    // ---- libstdc++ iteration order ----
    // Rounds > GS_SMALL_LAST are spread grid-wide.  (Keeping ALL rounds in the single workgroup per element saves ~50 launches
    // per fragment but was measured slower end to end, 580 vs 640 fragments/s: the serial rounds of the 30 k-voxel stage sit
    // on every fragment's critical path and four fragments in flight do not hide them.)
    const int small_last = GS_SMALL_LAST;
    gs_order_small_kernel<<<B, GS_SMALL_T, 0, stream>>>(A, small_last);
    for (int j = small_last + 1; j < D3F_NCHAIN && (long long)D3F_CHAIN_HOST[j - 1] < (long long)maxM; ++j) {
        const long long nbj = (long long)D3F_CHAIN_HOST[j];
        const int hi = (int)((long long)maxM < nbj ? (long long)maxM : nbj);
        // the place kernel of a round also inserts the next round (when there is one): its grid covers that round's positions
        const bool more = j + 1 < D3F_NCHAIN && nbj < (long long)maxM;
        const long long nbn = more ? (long long)D3F_CHAIN_HOST[j + 1] : 0;
        const int hin = more ? (int)((long long)maxM < nbn ? (long long)maxM : nbn) : hi;
        dim3 g(d3f_cdiv(hi, 256), B), gt(d3f_cdiv(hi, GS_TILE), B), gp(d3f_cdiv(hin > hi ? hin : hi, 256), B);
        if (j == small_last + 1) gs_order_insert_kernel<<<g, 256, 0, stream>>>(A, j);
        gs_order_scan_tiles_kernel<<<gt, 256, 0, stream>>>(A, j, B, counters + 4 + j * B);
        gs_order_place_kernel<<<gp, 256, 0, stream>>>(A, j);
    }
    if (use_sort)
        gs_emit_kernel<<<d3f_cdiv(M, 256), 256, 0, stream>>>(meta, moffs, B, A.vpos, bary, sub_points, M_cap);
    else
        gs_accum_kernel<<<d3f_cdiv(async ? N : M, 256), 256, 0, stream>>>(points, features, fdim, meta, moffs, B, vstart, sbase,
                                                                         vcnt, sorted, A.vpos, sub_points, sub_features, M_cap);
    if (ldim > 0) {
        const size_t tot = (size_t)M * (size_t)ldim;
        gs_fill_kernel<<<d3f_cdiv((long long)tot, 256), 256, 0, stream>>>(sub_classes, tot, (int)0x80000000);
        gs_labels_kernel<<<nblk, 256, 0, stream>>>(N, ldim, classes, pvid, moffs, B, A.vpos, sub_classes);
    }
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" int d3f_batch_grid_subsample(const float* points, int N, const int* lens_dev, int B, float dl,
                                        const float* features, int fdim, const int* classes, int ldim,
                                        float* sub_points, float* sub_features, int* sub_classes, int* sub_lens_dev,
                                        int* status_host, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || N > (1 << 30) || B < 1 || B > D3F_MAX_BATCH || !(dl > 0.f) || fdim < 0 || ldim < 0) return D3F_ERR_ARG;
    if (!points || !lens_dev || !sub_points || !sub_lens_dev || !status_host) return D3F_ERR_ARG;
    if ((fdim > 0 && (!features || !sub_features)) || (ldim > 0 && (!classes || !sub_classes))) return D3F_ERR_ARG;
    for (int i = 0; i < B + 2; ++i) status_host[i] = 0;
    if (N == 0) return d3f_fill_u32(sub_lens_dev, B, 0u, stream);
    return gs_run(points, N, lens_dev, B, dl, features, fdim, classes, ldim, sub_points, N, N, 0, sub_features, sub_classes,
                  sub_lens_dev, status_host, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int d3f_batch_grid_subsample_async(const float* points, int N_cap, const int* lens_dev, int B, float dl,
                                              float* sub_points, int M_cap, int elem_cap, int elem_points_cap, int* sub_lens_dev,
                                              int* status_dev, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N_cap < 1 || N_cap > (1 << 30) || M_cap < 1 || elem_cap < 0 || elem_points_cap < 0 || B < 1 || B > D3F_MAX_BATCH || !(dl > 0.f))
        return D3F_ERR_ARG;
    if (!points || !lens_dev || !sub_points || !sub_lens_dev || !status_dev) return D3F_ERR_ARG;
    return gs_run(points, N_cap, lens_dev, B, dl, nullptr, 0, nullptr, 0, sub_points, M_cap, elem_cap, elem_points_cap, nullptr,
                  nullptr, sub_lens_dev, nullptr, status_dev, workspace, workspace_bytes, stream);
}

// The same call with the clouds read IN PLACE: cloud b is its own array clouds_dev[b] (f32[lens[b], 3], any address), nothing is
// stacked.  N_cap bounds the SUM of the lengths (workspace and launch sizes).  What a replayed launch sequence needs to take its
// inputs where the producer left them: the pointer table and the lengths are the only per-replay uploads.
extern "C" int d3f_batch_grid_subsample_async_inplace(const float* const* clouds_dev, int N_cap, const int* lens_dev, int B, float dl,
                                                      float* sub_points, int M_cap, int elem_cap, int* sub_lens_dev, int* status_dev,
                                                      void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N_cap < 1 || N_cap > (1 << 30) || M_cap < 1 || elem_cap < 0 || B < 1 || B > D3F_MAX_BATCH || !(dl > 0.f)) return D3F_ERR_ARG;
    if (!clouds_dev || !lens_dev || !sub_points || !sub_lens_dev || !status_dev) return D3F_ERR_ARG;
    return gs_run(nullptr, N_cap, lens_dev, B, dl, nullptr, 0, nullptr, 0, sub_points, M_cap, elem_cap, 0, nullptr, nullptr, sub_lens_dev,
                  nullptr, status_dev, workspace, workspace_bytes, stream, clouds_dev);
}

// np.concatenate([pts, pts]) of the reference's test generators (datasets/ThreeDMatch.py:190-192, demo_registration.py:
// 72-79: every fragment is fed stacked with itself), for B clouds at once and with the row counts read from HBM:
// cloud b (rows [o_b, o_b + m_b) of pts) becomes rows [2 o_b, 2 o_b + m_b) and [2 o_b + m_b, 2 o_b + 2 m_b) of out,
// lens_out = [m_0, m_0, m_1, m_1, ...], total = 2 sum m_b.  B = 1 is the single self-pair.
__global__ void __launch_bounds__(256) gs_stack_pair_kernel(const float* __restrict__ pts, int M_cap, const int* __restrict__ lens_in,
                                                            int B, float* __restrict__ out, int* __restrict__ lens_out,
                                                            int* __restrict__ total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int tot = 0;
    for (int b = 0; b < B; ++b) tot += lens_in[b];
    tot = min(tot, M_cap);
    if (i == 0) {
        int room = M_cap;
        for (int b = 0; b < B; ++b) {
            const int m = min(lens_in[b], room);
            room -= m;
            lens_out[2 * b] = m;
            lens_out[2 * b + 1] = m;
        }
        *total = 2 * tot;
    }
    if (i >= 3 * tot) return;
    const int pi = i / 3, c = i - 3 * pi;
    int start = 0, m = 0;
    for (int b = 0; b < B; ++b) {        // cloud of point pi (B is small)
        m = lens_in[b];
        if (pi < start + m || b == B - 1) break;
        start += m;
    }
    m = min(m, tot - start);
    const float v = pts[i];
    const size_t row = 2 * (size_t)start + (size_t)(pi - start);
    out[3 * row + c] = v;
    out[3 * (row + m) + c] = v;
}

extern "C" int d3f_stack_self_pair(const float* pts, int M_cap, const int* lens_in_dev, int B, float* out, int* lens_out_dev,
                                   int* total_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M_cap < 1 || B < 1 || 2 * B > D3F_MAX_BATCH || !pts || !lens_in_dev || !out || !lens_out_dev || !total_dev)
        return D3F_ERR_ARG;
    gs_stack_pair_kernel<<<d3f_cdiv(3ll * M_cap, 256), 256, 0, stream>>>(pts, M_cap, lens_in_dev, B, out, lens_out_dev, total_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
