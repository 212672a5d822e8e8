// Fixed-radius neighbour search on gfx950: uniform cell grid + 16 / 32 lanes (a quarter / half of a wavefront) per query.
//
// Reference: tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:211-332 (batch_nanoflann_neighbors, the active
// path: KD-tree radiusSearch, rows sorted by d2) and :125-208 (batch_ordered_neighbors: brute force, stable).
// Semantics kept bit-exactly:
//   r2 = radius*radius in fp32; d2 = (dx*dx + dy*dy) + dz*dz with dx = query - support, every operation
//   rounded to fp32 and never contracted to FMA; a support is a neighbour iff d2 < r2 (strict); only supports
//   of the query's own batch element; indices are global; rows ascending by (d2, index) -- which IS
//   batch_ordered_neighbors and equals the nanoflann path except inside runs of bit-equal d2, whose order
//   nanoflann leaves to an unstable sort; pad value = total number of supports.
//
// MI355X design (bound by dependent L2 round trips, not by bytes: SQ counters show 75 % of wave cycles in s_waitcnt):
//   build (5 launches): reset -> bounding boxes, the LAST workgroup derives the grid geometry (cell edge slightly > radius
//           so the 27-cell stencil is a guaranteed superset; fp64 index arithmetic) -> atomic histogram -> one-launch
//           scan -> scatter into a float4 {x,y,z,index-bits} array: one 16-byte load per candidate, x-adjacent cells are
//           contiguous, so the 27-cell stencil is 9 contiguous runs.  The sorted index array doubles as a spatially
//           coherent visiting order for every per-point kernel of the level;
//   search (1 launch): LPQ = 32 lanes per query (two queries per wavefront; 16 lanes = four queries per wavefront for launches of
//           >= 100 k queries, where the rounds of resident wavefronts dominate) walk the 9 runs as one virtual list, four (eight)
//           candidate loads in flight per lane; hits are compacted into the group's LDS segment with a ballot + popcount
//           prefix (variable-length lists); the <= cap hits are ordered by rank counting (each lane counts how many hits
//           precede its own, four per 128-bit LDS read -- keys are unique so ranks are a permutation) and the first `width`
//           ranks are stored straight into the output row.  first_only keeps a running (d2, index) minimum instead.
//   Sizes come from the device lens arrays (capacity mode): Nq / Ns only bound grids and buffers.
#include "prims.h"
#include <cstdlib>

struct NbElem {
    double mn[3];
    double inv_h;
    int dims[3];
    int cbase;  // first cell of this element in the global cell array
};

#define NB_WAVES_PER_BLOCK 4
#define NB_EL_LDS 40      // batch elements whose grid geometry the search kernel copies to LDS (56 bytes each; 16 fragments = 32 clouds)

// one thread per element: choose the cell edge (>= radius*(1+2^-20); doubled until the element's grid fits
// its share of the cell budget), grid dims and cell base.
__device__ __forceinline__ void nb_prep(const unsigned* __restrict__ bbox, const int* __restrict__ soffs, int B, float radius,
                                        long long cell_budget, NbElem* __restrict__ el, int* __restrict__ ncells_total) {
    if (threadIdx.x != 0) return;
    long long base = 0;
    const long long per = cell_budget / B;
    for (int b = 0; b < B; ++b) {
        NbElem e;
        const int len = soffs[b + 1] - soffs[b];
        if (len <= 0) {
            e.mn[0] = e.mn[1] = e.mn[2] = 0.0;
            e.inv_h = 1.0;
            e.dims[0] = e.dims[1] = e.dims[2] = 1;
        } else {
            double mx[3];
            for (int d = 0; d < 3; ++d) {
                e.mn[d] = (double)d3f_ord2f(__hip_atomic_load(&bbox[b * 6 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                mx[d] = (double)d3f_ord2f(__hip_atomic_load(&bbox[b * 6 + 3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            double h = (double)radius * (1.0 + 1.0 / 1048576.0);
            if (!(h > 0.0)) h = 1.0;
            bool fits = false;
            for (int it = 0; it < 64; ++it) {
                // dims use the same expression as nb_cell_of, so every support's cell is in range by monotonicity
                e.inv_h = 1.0 / h;
                double tot = 1.0;
                for (int d = 0; d < 3; ++d) {
                    double n = floor((mx[d] - e.mn[d]) * e.inv_h) + 1.0;
                    e.dims[d] = (n < 1073741824.0) ? (int)n : 1073741824;
                    tot *= n;
                }
                if (tot <= (double)per) { fits = true; break; }
                h *= 2.0;
            }
            if (!fits) {
                // non-finite coordinates (never produced by the pipeline; possible when a flagged, overflowing capacity-mode
                // call left garbage rows upstream): one cell, every support a candidate -- slow but memory-safe
                e.mn[0] = e.mn[1] = e.mn[2] = 0.0;
                e.inv_h = 0.0;
                e.dims[0] = e.dims[1] = e.dims[2] = 1;
            }
        }
        e.cbase = (int)base;
        base += (long long)e.dims[0] * e.dims[1] * e.dims[2];
        el[b] = e;
    }
    ncells_total[0] = (int)base;
    ncells_total[1] = (int)base + 1;   // items of the cell scan: the searches also read the prefix one past the last cell
}

// epilogue of the bounding-box kernel: the last workgroup derives the grid geometry from the finished boxes
struct NbPrepEpi {
    const unsigned* bbox; const int* soffs; int B; float radius; long long cell_budget; NbElem* el; int* ncells;
    __device__ __forceinline__ void operator()() const { nb_prep(bbox, soffs, B, radius, cell_budget, el, ncells); }
};

__device__ __forceinline__ void nb_cell_of(const NbElem& e, float x, float y, float z, int& cx, int& cy, int& cz) {
    // monotone in each coordinate; (double)x - mn is exact for fp32 inputs of comparable magnitude
    cx = (int)floor(((double)x - e.mn[0]) * e.inv_h);
    cy = (int)floor(((double)y - e.mn[1]) * e.inv_h);
    cz = (int)floor(((double)z - e.mn[2]) * e.inv_h);
}

__global__ void __launch_bounds__(256) nb_count_kernel(const float* __restrict__ s, int Ns, const int* __restrict__ soffs,
                                                       int B, const NbElem* __restrict__ el, int* __restrict__ cell_of,
                                                       int* __restrict__ cell_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(Ns, soffs[B])) return;   // Ns is the capacity, soffs[B] the real number of supports
    const int b = d3f_find_elem(soffs, B, i);
    const NbElem e = el[b];
    int cx, cy, cz;
    nb_cell_of(e, s[3 * (size_t)i], s[3 * (size_t)i + 1], s[3 * (size_t)i + 2], cx, cy, cz);
    cx = min(max(cx, 0), e.dims[0] - 1);
    cy = min(max(cy, 0), e.dims[1] - 1);
    cz = min(max(cz, 0), e.dims[2] - 1);
    const int c = e.cbase + cx + e.dims[0] * (cy + e.dims[1] * cz);
    cell_of[i] = c;
    atomicAdd(&cell_cnt[c], 1);
}

__global__ void __launch_bounds__(256) nb_scatter_kernel(const float* __restrict__ s, int Ns, const int* __restrict__ ns_dev,
                                                         const int* __restrict__ cell_of,
                                                         const int* __restrict__ cell_start, const int* __restrict__ cell_base,
                                                         int* __restrict__ cell_cur,
                                                         float4* __restrict__ sorted, int* __restrict__ order,
                                                         int* __restrict__ inv, float* __restrict__ xyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(Ns, *ns_dev)) return;
    const int c = cell_of[i];
    const int pos = d3f_scan_at(cell_start, cell_base, c) + atomicAdd(&cell_cur[c], 1);
    const float x = s[3 * (size_t)i], y = s[3 * (size_t)i + 1], z = s[3 * (size_t)i + 2];
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
    order[pos] = i;
    inv[i] = pos;
    xyz[3 * (size_t)pos] = x; xyz[3 * (size_t)pos + 1] = y; xyz[3 * (size_t)pos + 2] = z;
}

// ---- search: one wavefront per query -----------------------------------------------------------------
// Latency is the enemy here (each query touches ~100 candidates in 9 runs): the 18 run bounds are fetched by 18
// lanes in ONE round trip, the 9 runs are then walked as a single virtual list (lane -> run by comparing against
// the wave-uniform run prefix), so a typical query needs two 64-wide candidate loads instead of 9+ dependent steps.
// LPQ lanes cooperate on one query (64 = one wavefront per query; 32 / 16 = two / four queries per wavefront).  A typical
// query has ~120 candidates and ~40 hits: with a whole wavefront per query half the lanes idle in every phase and the kernel
// is bound by the dependent memory round trips per wavefront, so packing several queries into a wavefront raises the work
// in flight per round trip.  Lane groups are independent: shuffles use width LPQ, ballots are masked to the group.
// ---- ordering of a query's hits: bitonic network over 64 keys, two per lane of a 32-lane group ----------------------------
// The rows must come out ascending by (d2, index).  Counting ranks against every other hit costs m^2 comparisons fed by LDS
// reads (m ~ 40-70: the larger half of this kernel's time); the keys (d2 bits << 32 | index: d2 >= +0, so its bit pattern
// orders like the value; indices are unique) instead go through the 21 compare-exchange stages of a 64-key bitonic sort, in
// registers: partners at distance 1 / 2 via DPP quad permutes, 4 / 8 / 16 via ds_swizzle (no LDS memory), distance 32 is the
// lane's own second key.  Element e = 32*slot + lane ends up holding rank e.
template <int J>
__device__ __forceinline__ unsigned nb_xor_lane(unsigned v) {
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1F);                          // lane ^ J (bit mode)
}
template <int J>
__device__ __forceinline__ void nb_cmpx(unsigned& hi, unsigned& lo, bool take_min) {
    const unsigned phi = nb_xor_lane<J>(hi), plo = nb_xor_lane<J>(lo);
    const bool p_less = (phi < hi) || (phi == hi && plo < lo);
    const bool take_p = (take_min == p_less);       // keys are unique (or both the +inf padding): !p_less means p is greater
    hi = take_p ? phi : hi;
    lo = take_p ? plo : lo;
}
// lane: 0..31 inside the group; (h0,l0) = element lane, (h1,l1) = element 32 + lane
__device__ __forceinline__ void nb_bitonic64(int lane, unsigned& h0, unsigned& l0, unsigned& h1, unsigned& l1) {
#define NB_STAGE(K_, J_)                                                                         \
    do {                                                                                         \
        const bool low_ = (lane & (J_)) == 0;                                                    \
        const bool up0_ = ((K_) >= 32) ? true : ((lane & (K_)) == 0);                            \
        const bool up1_ = ((K_) == 64) ? true : (((K_) == 32) ? false : ((lane & (K_)) == 0));   \
        nb_cmpx<J_>(h0, l0, up0_ == low_);                                                       \
        nb_cmpx<J_>(h1, l1, up1_ == low_);                                                       \
    } while (0)
    NB_STAGE(2, 1);
    NB_STAGE(4, 2); NB_STAGE(4, 1);
    NB_STAGE(8, 4); NB_STAGE(8, 2); NB_STAGE(8, 1);
    NB_STAGE(16, 8); NB_STAGE(16, 4); NB_STAGE(16, 2); NB_STAGE(16, 1);
    NB_STAGE(32, 16); NB_STAGE(32, 8); NB_STAGE(32, 4); NB_STAGE(32, 2); NB_STAGE(32, 1);
    {   // K = 64, J = 32: the partner is the lane's own other key; ascending for both
        const bool swap = (h1 < h0) || (h1 == h0 && l1 < l0);
        const unsigned th = swap ? h1 : h0, tl = swap ? l1 : l0;
        h1 = swap ? h0 : h1; l1 = swap ? l0 : l1;
        h0 = th; l0 = tl;
    }
    NB_STAGE(64, 16); NB_STAGE(64, 8); NB_STAGE(64, 4); NB_STAGE(64, 2); NB_STAGE(64, 1);
#undef NB_STAGE
}

// The same 64-key network for a 16-lane group, four keys per lane: element e = 16 * slot + lane.  Partners at distance 1 / 2 via
// DPP, 4 / 8 via ds_swizzle, 16 / 32 are the lane's own other slots (register compare-exchange, no cross-lane traffic).
__device__ __forceinline__ void nb_cmpx_own(unsigned& ha, unsigned& la, unsigned& hb, unsigned& lb, bool ascending) {
    const bool b_less = (hb < ha) || (hb == ha && lb < la);
    const bool swap = (b_less == ascending);        // ascending: the smaller key ends in a
    const unsigned th = swap ? hb : ha, tl = swap ? lb : la;
    hb = swap ? ha : hb; lb = swap ? la : lb;
    ha = th; la = tl;
}
__device__ __forceinline__ void nb_bitonic64x16(int lane, unsigned (&h)[4], unsigned (&l)[4]) {
    // direction of element e in the merge of size K: ascending iff (e & K) == 0 (K = 64: always)
#define NB_UP(K_, S_) (((K_) == 64) ? true : ((K_) >= 16 ? ((((S_) * 16) & (K_)) == 0) : ((lane & (K_)) == 0)))
#define NB_XSTAGE(K_, J_)                                                                        \
    do {                                                                                         \
        const bool low_ = (lane & (J_)) == 0;                                                    \
        nb_cmpx<J_>(h[0], l[0], NB_UP(K_, 0) == low_);                                           \
        nb_cmpx<J_>(h[1], l[1], NB_UP(K_, 1) == low_);                                           \
        nb_cmpx<J_>(h[2], l[2], NB_UP(K_, 2) == low_);                                           \
        nb_cmpx<J_>(h[3], l[3], NB_UP(K_, 3) == low_);                                           \
    } while (0)
    NB_XSTAGE(2, 1);
    NB_XSTAGE(4, 2); NB_XSTAGE(4, 1);
    NB_XSTAGE(8, 4); NB_XSTAGE(8, 2); NB_XSTAGE(8, 1);
    NB_XSTAGE(16, 8); NB_XSTAGE(16, 4); NB_XSTAGE(16, 2); NB_XSTAGE(16, 1);
    // K = 32: distance 16 = slots (0,1) ascending, (2,3) descending
    nb_cmpx_own(h[0], l[0], h[1], l[1], true);
    nb_cmpx_own(h[2], l[2], h[3], l[3], false);
    NB_XSTAGE(32, 8); NB_XSTAGE(32, 4); NB_XSTAGE(32, 2); NB_XSTAGE(32, 1);
    // K = 64: distance 32 = slots (0,2), (1,3); distance 16 = (0,1), (2,3); all ascending
    nb_cmpx_own(h[0], l[0], h[2], l[2], true);
    nb_cmpx_own(h[1], l[1], h[3], l[3], true);
    nb_cmpx_own(h[0], l[0], h[1], l[1], true);
    nb_cmpx_own(h[2], l[2], h[3], l[3], true);
    NB_XSTAGE(64, 8); NB_XSTAGE(64, 4); NB_XSTAGE(64, 2); NB_XSTAGE(64, 1);
#undef NB_XSTAGE
#undef NB_UP
}

template <bool FIRST_ONLY, int LPQ, bool HINT>
__device__ __forceinline__ void
nb_search_body(const float* __restrict__ q, int Nq, const int* __restrict__ qlens, int B,
               const NbElem* __restrict__ el, const int* __restrict__ cell_start, const int* __restrict__ cell_base,
               const float4* __restrict__ sorted,
               const float4* __restrict__ qorder, float r2, int pad, const int* __restrict__ ns_dev, int* __restrict__ out,
               int ld, int width, int cap, int* __restrict__ status, int want_kmax, float nn_hint, const int* __restrict__ inv) {
    // inv != NULL: INTERNAL numbering -- row wq (the visit position) instead of row qi, entries = inv[index]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QPB = 64 * NB_WAVES_PER_BLOCK / LPQ;   // queries per workgroup
    const int grp = threadIdx.x / LPQ, lane = threadIdx.x % LPQ;
    const int gshift = (threadIdx.x & 63) - lane;         // first lane of this group inside its wavefront
    const unsigned long long gmask = (LPQ == 64) ? ~0ull : (((1ull << LPQ) - 1ull) << gshift);
    float* hd2 = (float*)smem + (size_t)grp * 2 * cap;
    int* hidx = (int*)hd2 + cap;
    // the per-element grid geometry (a handful of 56-byte records) is staged in LDS by the workgroup's first lanes while the
    // queries' own first loads are in flight: el[b] then costs an LDS read instead of a dependent memory round trip
    __shared__ NbElem sel[NB_EL_LDS];
    if (B <= NB_EL_LDS) {
        constexpr int W = (int)(sizeof(NbElem) / sizeof(int));
        for (int i = threadIdx.x; i < B * W; i += blockDim.x) ((int*)sel)[i] = ((const int*)el)[i];
        __syncthreads();
    }
    int nq_real = 0;
    for (int j = 0; j < B; ++j) nq_real += qlens[j];
    const int nq = min(Nq, nq_real);      // Nq is the capacity, sum(qlens) the real number of queries
    if ((int)(blockIdx.x * QPB) >= nq) return;
    // one contiguous run of query blocks per XCD (common.h): neighbouring blocks read the same candidate runs
    const int wq = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((nq + QPB - 1) / QPB)) * QPB + grp;
    if (wq >= nq) return;
    if (pad == D3F_PAD_NUM_SUPPORTS) pad = *ns_dev;
    // queries that ARE the supports are visited in cell order: neighbouring groups then share their candidate runs in L2;
    // the cell-sorted copy holds position AND index of the wq-th support in one 16-byte record, so the query needs no
    // (order -> point) chain of dependent loads: one round trip less on the kernel's critical path
    int qi = wq;
    float qx, qy, qz;
    if (qorder) {               // the cell-sorted records of the grid the queries were sorted by (their own, or this one)
        const float4 me = qorder[wq];
        qi = __float_as_int(me.w);
        qx = me.x; qy = me.y; qz = me.z;
    } else {
        qx = q[3 * (size_t)qi]; qy = q[3 * (size_t)qi + 1]; qz = q[3 * (size_t)qi + 2];
    }
    int b = 0;   // batch element of the query: the last one starting at or before qi (lens -> offsets on the fly, B is small)
    for (int j = 1, start = qlens[0]; j < B; ++j) { if (qi >= start) b = j; start += qlens[j]; }
    const NbElem e = (B <= NB_EL_LDS) ? sel[b] : el[b];
    int cx, cy, cz;
    nb_cell_of(e, qx, qy, qz, cx, cy, cz);
    cx = min(max(cx, -2), e.dims[0] + 1);
    cy = min(max(cy, -2), e.dims[1] + 1);
    cz = min(max(cz, -2), e.dims[2] + 1);
    // Stencil: the 3 x 3 x 3 cells around the query's.  FIRST_ONLY with nn_hint > 0 (the caller expects the nearest support
    // within that distance -- the upsampling matrices: a point's own voxel barycentre is at most sqrt(3) dl away): the first
    // attempt only visits the cells that the ball of radius nn_hint touches (8 of 27 when nn_hint < cell / 2 ... 0.7 cell) and
    // is accepted when it finds a support within nn_hint -- then nothing closer can lie outside those cells.  Otherwise the
    // full stencil is searched: the result never depends on the hint.
    int n = 0;  // group-uniform hit count
    const unsigned long long lt = (1ull << lane) - 1ull;
    float bd2 = 3.4e38f;
    int bidx = 0x7fffffff;
    auto scan = [&](bool restricted) {
    int xl = cx - 1, xh = cx + 1, yl = cy - 1, yh = cy + 1, zl = cz - 1, zh = cz + 1;
    if (restricted) {
        // cells touched by the ball, in the grid's own fp64 index arithmetic (monotone, so a support within nn_hint of the
        // query lies in [lo, hi] on every axis)
        const double h = (double)nn_hint;
        xl = max(xl, (int)floor(((double)qx - h - e.mn[0]) * e.inv_h)); xh = min(xh, (int)floor(((double)qx + h - e.mn[0]) * e.inv_h));
        yl = max(yl, (int)floor(((double)qy - h - e.mn[1]) * e.inv_h)); yh = min(yh, (int)floor(((double)qy + h - e.mn[1]) * e.inv_h));
        zl = max(zl, (int)floor(((double)qz - h - e.mn[2]) * e.inv_h)); zh = min(zh, (int)floor(((double)qz + h - e.mn[2]) * e.inv_h));
    }
    const int x0 = max(xl, 0), x1 = min(xh, e.dims[0] - 1);
    // lanes 0..8: start and end of the 9 (y,z) rows of the stencil (both fetched by the run's lane: one round trip, and the
    // 16-lane form of the kernel has no lanes 9..17 to hold the ends)
    int bound = 0, len_l = 0;
    if (lane < 9) {
        const int j = lane;
        const int y = cy + (j % 3) - 1, z = cz + (j / 3) - 1;
        if (x0 <= x1 && y >= max(yl, 0) && y <= min(yh, e.dims[1] - 1) && z >= max(zl, 0) && z <= min(zh, e.dims[2] - 1)) {
            const int rowbase = e.cbase + e.dims[0] * (y + e.dims[1] * z);
            bound = d3f_scan_at(cell_start, cell_base, rowbase + x0);
            len_l = d3f_scan_at(cell_start, cell_base, rowbase + x1 + 1) - bound;
        }
    }
    static_assert(LPQ >= 16, "the 9 runs of the stencil live in one lane group");
    // group-uniform run offsets: off[j] = lo[j] - prefix[j]; prefix kept for the lane -> run test
    int pre[10], off[9];
    pre[0] = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int lo_j = __shfl(bound, j, LPQ);
        const int ln_j = __shfl(len_l, j, LPQ);
        off[j] = lo_j - pre[j];
        pre[j + 1] = pre[j] + ln_j;
    }
    const int T = pre[9];
    n = 0;
    bd2 = 3.4e38f;
    bidx = 0x7fffffff;
    // four candidate loads in flight per lane (a typical query's ~120 candidates in ONE round trip): the loads of consecutive
    // steps are independent, only the hit compaction is sequential.  Straight-line loads: a lane beyond the list re-reads
    // the list's first candidate (address clamp) instead of branching around the load.
    constexpr int NLD = (LPQ == 16 && !FIRST_ONLY) ? 8 : 4;   // (16 lanes per query: the same ~128 candidates per round trip;
                                                               //  the nearest-only scan visits 8 cells, ~35 candidates)
    for (int v0 = 0; v0 < T; v0 += NLD * LPQ) {
        float4 sp[NLD];
        bool in[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int vv = v0 + u * LPQ + lane;
            in[u] = vv < T;
            const int v = in[u] ? vv : 0;
            int t = v + off[0];
#pragma unroll
            for (int j = 1; j < 9; ++j) t = (v >= pre[j]) ? v + off[j] : t;
            sp[u] = sorted[t];
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            if (v0 + u * LPQ >= T) break;   // group-uniform
            const float dx = __fsub_rn(qx, sp[u].x), dy = __fsub_rn(qy, sp[u].y), dz = __fsub_rn(qz, sp[u].z);
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            const int si = __float_as_int(sp[u].w);
            const bool hit = in[u] && d2 < r2;
            const unsigned long long m = (__ballot(hit) & gmask) >> gshift;   // hits of this group, bit = lane in group
            if (FIRST_ONLY) {
                if (hit && (d2 < bd2 || (d2 == bd2 && si < bidx))) { bd2 = d2; bidx = si; }
            } else if (hit) {
                const int pos = n + __popcll(m & lt);
                if (pos < cap) { hd2[pos] = d2; hidx[pos] = si; }
            }
            n += __popcll(m);
        }
    }
    if (FIRST_ONLY) {
        // lexicographic (d2, index) minimum over the group
#pragma unroll
        for (int o = LPQ / 2; o > 0; o >>= 1) {
            const float od = __shfl_xor(bd2, o, LPQ);
            const int oi = __shfl_xor(bidx, o, LPQ);
            if (od < bd2 || (od == bd2 && oi < bidx)) { bd2 = od; bidx = oi; }
        }
    }
    };
    if (FIRST_ONLY && HINT) {
        scan(true);
        // (group-uniform) the restricted scan stands when its nearest support lies within the hint (0.998: the fp32 d2 of a
        // support just outside the visited cells can round below that of one just inside the ball); else: the full stencil
        if (!(n > 0 && bd2 <= 0.998f * nn_hint * nn_hint)) scan(false);
    } else {
        scan(false);
    }
    if (lane == 0) {
        // one shared word: an unconditional atomic per query serialises at ~12 ns each in L2 (60k queries = 0.7 ms);
        // read first, update only when this query raises the maximum (a handful of times per launch)
        if (want_kmax && n > __hip_atomic_load(&status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&status[0], n);
        if (!FIRST_ONLY && n > cap) atomicOr(&status[1], D3F_ST_HIT_OVERFLOW);
    }
    int* row = out + (size_t)(inv ? wq : qi) * ld;
    if (FIRST_ONLY) {
        if (lane == 0 && width > 0) row[0] = (n > 0) ? (inv ? inv[bidx] : bidx) : pad;
        for (int j = 1 + lane; j < width; j += LPQ) row[j] = pad;
        return;
    }
    const int m = min(n, cap);
    // rank counting, four hits per LDS read (ds_read_b128 broadcasts): pad the tail of the last quad with +inf keys
    const int m4 = (m + 3) & ~3;
    if (m + lane < m4) { hd2[m + lane] = 3.4e38f; hidx[m + lane] = 0x7fffffff; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (LPQ == 16 && n <= 64 && cap >= 64) {      // group-uniform: the common case goes through the register network
        unsigned h[4], l4[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            h[sl] = 0xFFFFFFFFu; l4[sl] = 0xFFFFFFFFu;
            if (lane + 16 * sl < m) { h[sl] = __float_as_uint(hd2[lane + 16 * sl]); l4[sl] = (unsigned)hidx[lane + 16 * sl]; }
        }
        nb_bitonic64x16(lane, h, l4);
        const int mw = min(m, width);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (lane + 16 * sl < mw) row[lane + 16 * sl] = inv ? inv[(int)l4[sl]] : (int)l4[sl];
        for (int j = m + lane; j < width; j += LPQ) row[j] = pad;
        return;
    }
    if (LPQ == 32 && n <= 64 && cap >= 64) {      // group-uniform: the common case goes through the register network
        unsigned h0 = 0xFFFFFFFFu, l0 = 0xFFFFFFFFu, h1 = 0xFFFFFFFFu, l1 = 0xFFFFFFFFu;
        if (lane < m) { h0 = __float_as_uint(hd2[lane]); l0 = (unsigned)hidx[lane]; }
        if (lane + 32 < m) { h1 = __float_as_uint(hd2[lane + 32]); l1 = (unsigned)hidx[lane + 32]; }
        nb_bitonic64(lane, h0, l0, h1, l1);
        const int mw = min(m, width);
        if (lane < mw) row[lane] = inv ? inv[(int)l0] : (int)l0;
        if (lane + 32 < mw) row[lane + 32] = inv ? inv[(int)l1] : (int)l1;
        for (int j = m + lane; j < width; j += LPQ) row[j] = pad;
        return;
    }
    for (int e0 = 0; e0 < m; e0 += LPQ) {
        const int ei = e0 + lane;
        if (ei < m) {
            const float kd = hd2[ei];
            const int ki = hidx[ei];
            int rank = 0;
            for (int j = 0; j < m4; j += 4) {
                const float4 dj = *(const float4*)&hd2[j];
                const int4 ij = *(const int4*)&hidx[j];
                rank += (dj.x < kd || (dj.x == kd && ij.x < ki)) ? 1 : 0;
                rank += (dj.y < kd || (dj.y == kd && ij.y < ki)) ? 1 : 0;
                rank += (dj.z < kd || (dj.z == kd && ij.z < ki)) ? 1 : 0;
                rank += (dj.w < kd || (dj.w == kd && ij.w < ki)) ? 1 : 0;
            }
            if (rank < width) row[rank] = inv ? inv[ki] : ki;
        }
    }
    for (int j = m + lane; j < width; j += LPQ) row[j] = pad;
}

template <bool FIRST_ONLY, int LPQ, bool HINT>
__global__ void __launch_bounds__(64 * NB_WAVES_PER_BLOCK)
nb_search_kernel(const float* __restrict__ q, int Nq, const int* __restrict__ qlens, int B,
                 const NbElem* __restrict__ el, const int* __restrict__ cell_start, const int* __restrict__ cell_base,
                 const float4* __restrict__ sorted,
                 const float4* __restrict__ qorder, float r2, int pad, const int* __restrict__ ns_dev, int* __restrict__ out,
                 int ld, int width, int cap, int* __restrict__ status, int want_kmax, float nn_hint, const int* __restrict__ inv) {
    nb_search_body<FIRST_ONLY, LPQ, HINT>(q, Nq, qlens, B, el, cell_start, cell_base, sorted, qorder, r2, pad, ns_dev, out, ld, width,
                                          cap, status, want_kmax, nn_hint, inv);
}

#include "nb_cell_search.h"
#include "nb_nearest.h"

// ------------------------------------------------------------------------------------------------
// cells the grid may use: 4 per support (surface clouds occupy ~0.3 cells per point at cell edge = radius; a sparser cloud gets
// larger cells).  The budget is what every build has to clear -- in capacity mode whatever the real cloud size -- so it is
// kept tight: 16 per support meant 40 MB of fills per level-0 build.
static long long nb_cell_budget(int Ns) {
    long long b = 4ll * (long long)(Ns > 0 ? Ns : 1);
    if (b < (1ll << 16)) b = 1ll << 16;
    if (b > (1ll << 28)) b = 1ll << 28;
    return b;
}

// The grid object is a plain arena inside caller-owned memory; build and search carve it identically.
struct NbGrid {
    int* soffs; unsigned* bbox; NbElem* el; int* ncells; unsigned* counters; int* cell_cnt; int* cell_cur; int* cell_start; int* cell_of;
    float4* sorted; int* order; int* stmp;
    int* inv;          // inverse of `order`: position of support i in the cell-sorted arrays (the INTERNAL numbering of the level)
    float* xyz;        // f32[Ns, 3]: the supports in cell-sorted order (what a model running on the internal numbering reads)
    long long cells;
    bool ok;
};
static NbGrid nb_carve(void* ws, size_t bytes, int Ns, int B) {
    NbGrid g;
    g.cells = nb_cell_budget(Ns) + 8;
    D3fArena ar(ws, bytes);
    const size_t ns = (size_t)(Ns > 0 ? Ns : 1);
    g.soffs = ar.take<int>(B + 1);
    g.bbox = ar.take<unsigned>(B * 6);
    g.el = ar.take<NbElem>(B);
    g.ncells = ar.take<int>(16);
    g.counters = (unsigned*)(g.ncells + 8);   // two ticket counters behind the cell count
    g.cell_cnt = ar.take<int>((size_t)g.cells * 2);   // [counts | cursors]: one memset
    g.cell_cur = g.cell_cnt ? g.cell_cnt + g.cells : nullptr;
    g.cell_start = ar.take<int>((size_t)g.cells);
    g.cell_of = ar.take<int>(ns);
    g.sorted = ar.take<float4>(ns);
    g.order = ar.take<int>(ns);
    g.stmp = ar.take<int>(d3f_scan_base_ints((int)g.cells));
    g.inv = ar.take<int>(ns);
    g.xyz = ar.take<float>(3 * ns);
    g.ok = ar.ok;
    return g;
}

extern "C" size_t d3f_neighbor_grid_bytes(int Ns, int B) {
    if (Ns < 0 || B < 1) return 0;
    const long long cells = nb_cell_budget(Ns) + 8;
    const size_t ns = (size_t)(Ns > 0 ? Ns : 1);
    size_t bytes = 0;
    bytes += d3f_align((B + 1) * sizeof(int)) + d3f_align(B * 6 * sizeof(unsigned)) + d3f_align(B * sizeof(NbElem)) + d3f_align(64);
    bytes += d3f_align((size_t)cells * 2 * sizeof(int)) + d3f_align((size_t)cells * sizeof(int));
    bytes += 2 * d3f_align(ns * sizeof(int)) + d3f_align(ns * sizeof(float4));
    bytes += d3f_align(d3f_scan_base_ints((int)cells) * sizeof(int));
    bytes += d3f_align(ns * sizeof(int)) + d3f_align(3 * ns * sizeof(float));
    return bytes + 1024;
}

// byte offset, inside a built grid object, of `order`: i32[Ns] = support indices sorted by cell -- a spatially coherent
// processing order for any per-point kernel over the same cloud (neighbouring entries share most of their neighbours)
extern "C" size_t d3f_neighbor_grid_order_offset(int Ns, int B) {
    if (Ns < 0 || B < 1) return 0;
    NbGrid g = nb_carve((void*)0x1000, (size_t)1 << 60, Ns, B);
    return (size_t)((char*)g.order - (char*)0x1000);
}

// byte offsets of `inv` i32[Ns] (position of support i in cell order) and `xyz` f32[Ns, 3] (the supports in cell order)
extern "C" size_t d3f_neighbor_grid_inv_offset(int Ns, int B) {
    if (Ns < 0 || B < 1) return 0;
    NbGrid g = nb_carve((void*)0x1000, (size_t)1 << 60, Ns, B);
    return (size_t)((char*)g.inv - (char*)0x1000);
}
extern "C" size_t d3f_neighbor_grid_xyz_offset(int Ns, int B) {
    if (Ns < 0 || B < 1) return 0;
    NbGrid g = nb_carve((void*)0x1000, (size_t)1 << 60, Ns, B);
    return (size_t)((char*)g.xyz - (char*)0x1000);
}

extern "C" int d3f_neighbor_grid_build(const float* supports, int Ns, const int* s_lens_dev, int B, float radius,
                                       void* grid, size_t grid_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Ns < 0 || B < 1 || B > D3F_MAX_BATCH || !(radius >= 0.f) || !s_lens_dev || (Ns > 0 && !supports) || !grid)
        return D3F_ERR_ARG;
    NbGrid g = nb_carve(grid, grid_bytes, Ns, B);
    if (!g.ok) return D3F_ERR_WORKSPACE;
    int rc;
    // 5 launches: reset (offsets, boxes, tickets, cell counters) -> boxes (+ grid geometry in the last workgroup) ->
    // histogram -> one-launch scan -> scatter.  The cell arrays are sized for the whole budget; scanning all of it keeps
    // the launch shapes static (no host read-back of the real cell count).
    const D3fFill none{nullptr, 0ull, 0u};
    if ((rc = d3f_begin_launch(s_lens_dev, B, g.soffs, g.bbox, g.counters, 2,
                               D3fFill{(unsigned*)g.cell_cnt, (unsigned long long)g.cells * 2ull, 0u}, none, none, none,
                               stream)) != D3F_OK) return rc;
    NbPrepEpi prep{g.bbox, g.soffs, B, radius, nb_cell_budget(Ns), g.el, g.ncells};
    if ((rc = d3f_bbox_launch_t(supports, g.soffs, B, Ns, g.bbox, g.counters, prep, stream)) != D3F_OK) return rc;
    if (Ns > 0) {
        nb_count_kernel<<<d3f_cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, g.soffs, B, g.el, g.cell_of, g.cell_cnt);
        D3F_LAUNCH_CHECK();
    }
    if ((rc = d3f_scan_fold_launch(D3fScanIn{g.cell_cnt}, (int)g.cells, g.ncells + 1, g.cell_start, g.stmp, g.counters + 1, D3fNoEpi{},
                                   stream)) != D3F_OK) return rc;
    if (Ns > 0) {
        nb_scatter_kernel<<<d3f_cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, g.soffs + B, g.cell_of, g.cell_start, g.stmp,
                                                                 g.cell_cur, g.sorted, g.order, g.inv, g.xyz);
        D3F_LAUNCH_CHECK();
    }
    return D3F_OK;
}

// ---- the searches of a built grid: one dispatcher ---------------------------------------------------------------------------------
// qsorted: cell-sorted records {x, y, z, index} of a grid built over the QUERIES -- the visiting order (neighbouring wavefronts /
// lane groups then read the same support runs); the grid's own records when the queries ARE its supports; NULL: plain order.
// Three kernels, chosen by what is asked and by the size of the launch (profiles/r06_experiments.txt n10; us per launch at the
// engine's F = 4 shapes, stand-alone):
//   full rows, queries = supports   nb_cell_search_kernel (one wavefront per query, stencil shared by a cell, rows ordered eight
//                                   at a time in registers) from 40 k queries on (235 k: 102 -> 82) and below 6 k (the lane-group
//                                   kernel has a ~18 us floor there: 3.6 k: 18.8 -> 13.1, 0.8 k: 19.5 -> 6.8); in between the
//                                   launch is less than a round of wavefronts and the lane-group kernel's four queries per
//                                   wavefront keep more loads in flight (14.5 k: 20.7 against 28.0)
//   full rows, other queries        nb_search_kernel (16 / 32 lanes per query): without a shared stencil a wavefront per query only
//                                   adds latency (58 k pool queries: 35 against 53)
//   nearest only, no Kmax           nb_nearest_kernel (four lanes per query, a stencil row per lane) from 40 k queries on
//                                   (235 k: 59 -> 48 with the queries' own cell order, 58 k: 25 -> 20); nb_search_kernel below
// D3F_NB_CELL / D3F_NB_NEAREST = 0: never, = 2: always (A/B measurements; read per call so that one process can switch).
static int nb_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static int nb_search_dispatch(const NbGrid& g, const float* queries, int Nq, const int* q_lens_dev, int B, float radius,
                              const float4* qsorted, bool queries_are_supports, int* out, int ld, int width, int pad_value, int cap,
                              int first_only, float nn_hint, int want_kmax, int* status_dev, hipStream_t stream, bool internal = false) {
    const float r2 = radius * radius;
    // internal numbering (D3F_NB_INTERNAL): rows by visit position (needs a visiting order), entries through the grid's `inv`
    if (internal && !qsorted) return D3F_ERR_ARG;
    const int* inv = internal ? g.inv : nullptr;
    if (first_only && !want_kmax) {
        const int mode = nb_env("D3F_NB_NEAREST", 1);
        if (mode == 2 || (mode == 1 && Nq >= 40000)) {
            const int blocks = d3f_cdiv(Nq, 64);
            if (nn_hint > 0.f)
                nb_nearest_kernel<true><<<blocks, 256, 0, stream>>>(queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, qsorted,
                                                                    r2, pad_value, g.soffs + B, out, ld, width, nn_hint, inv);
            else
                nb_nearest_kernel<false><<<blocks, 256, 0, stream>>>(queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, qsorted,
                                                                     r2, pad_value, g.soffs + B, out, ld, width, nn_hint, inv);
            D3F_LAUNCH_CHECK();
            return D3F_OK;
        }
    }
    if (!first_only && queries_are_supports) {
        const int mode = nb_env("D3F_NB_CELL", 1);
        if (mode == 2 || (mode == 1 && (Nq >= 40000 || Nq < 6000))) {
            cap = (cap + 1) & ~1;
            // queries per wavefront: consecutive queries share a stencil, and eight rows are ordered at once -- but a wavefront works
            // through its queries one after the other: a small launch keeps fewer per wavefront (all of them in flight at once)
            const int q_forced = nb_env("D3F_NBC_Q", 0);
            int Q = q_forced > 0 ? q_forced : Nq / 2048;
            Q = Q < 1 ? 1 : (Q > 16 ? 16 : Q);
            if (Q > NBC_QMAX) Q = NBC_QMAX;
            const int dbg = nb_env("D3F_NBC_DBG", 0);         // measurement only: skip phases (results are then wrong)
            const char* prof_env = getenv("D3F_NBC_PROF");     // measurement only: device address of 8 u64 words per wavefront (hex)
            unsigned long long* prof = prof_env ? (unsigned long long*)strtoull(prof_env, nullptr, 16) : nullptr;
            const int blocks = d3f_cdiv(d3f_cdiv(Nq, Q), 4);
            const size_t lds = (size_t)4 * ((size_t)cap * 8 + NBC_BATCH * 64 * 8);
            if (inv)
                nb_cell_search_kernel<true, true><<<blocks, 256, lds, stream>>>(queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, r2,
                                                                                pad_value, g.soffs + B, out, ld, width, cap, status_dev, want_kmax, Q, dbg, prof, inv);
            else
                nb_cell_search_kernel<true, false><<<blocks, 256, lds, stream>>>(queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, r2,
                                                                                 pad_value, g.soffs + B, out, ld, width, cap, status_dev, want_kmax, Q, dbg, prof, inv);
            D3F_LAUNCH_CHECK();
            return D3F_OK;
        }
    }
    // lanes per query: 32 (two queries per wavefront) unless the ordering budget is large (rare, dense clouds)
    cap = (cap + 3) & ~3;   // LDS segments are read four hits at a time
    // lanes per query.  The kernel runs at full occupancy and costs (rounds of resident wavefronts) x (a chain of ~4 dependent
    // round trips): 16 lanes per query -- four queries per wavefront, eight candidate loads in flight per lane, four keys per lane
    // in the ordering network -- halves the rounds of a large launch (level 0 at F = 4: 111 -> 95 us, nearest-only 77 -> 63 us)
    // but lengthens the chain of a launch that is less than a round anyway (levels 2-4: 17 -> 21-29 us), so it is chosen by size
    // (profiles/r03_experiments.txt x19); 64 lanes only for the large ordering budget.
    const int lpq = cap > 256 ? 64 : (Nq >= 100000 ? 16 : 32);
    const int qpb = 64 * NB_WAVES_PER_BLOCK / lpq;
    const int blocks = d3f_cdiv(Nq, qpb);
    const size_t lds = first_only ? 0 : (size_t)qpb * cap * 2 * sizeof(float);
    const int kcap = first_only ? 1 : cap;
#define D3F_NB(FO_, LPQ_)                                                                                              \
    if (FO_ && nn_hint > 0.f)                                                                                          \
        nb_search_kernel<FO_, LPQ_, FO_><<<blocks, 64 * NB_WAVES_PER_BLOCK, lds, stream>>>(                             \
            queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, qsorted, r2, pad_value, g.soffs + B, out, ld, width, \
            kcap, status_dev, want_kmax, nn_hint, inv);                                                                \
    else                                                                                                               \
    nb_search_kernel<FO_, LPQ_, false><<<blocks, 64 * NB_WAVES_PER_BLOCK, lds, stream>>>(                               \
        queries, Nq, q_lens_dev, B, g.el, g.cell_start, g.stmp, g.sorted, qsorted, r2, pad_value, g.soffs + B, out, ld, width, \
        kcap, status_dev, want_kmax, nn_hint, inv)
    if (first_only) { if (lpq == 64) D3F_NB(true, 64); else if (lpq == 32) D3F_NB(true, 32); else D3F_NB(true, 16); }
    else { if (lpq == 64) D3F_NB(false, 64); else if (lpq == 32) D3F_NB(false, 32); else D3F_NB(false, 16); }
#undef D3F_NB
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" int d3f_neighbor_grid_search_ordered(const void* grid, size_t grid_bytes, int Ns, const float* queries, int Nq,
                                                const int* q_lens_dev, int B, float radius, const void* query_grid,
                                                size_t query_grid_bytes, int* out, int ld, int width, int pad_value, int cap,
                                                int first_only, float nn_hint, int flags, int* status_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Nq < 0 || Ns < 0 || B < 1 || B > D3F_MAX_BATCH || width < 0 || ld < width || !(radius >= 0.f)) return D3F_ERR_ARG;
    if (!(nn_hint >= 0.f) || nn_hint >= radius) nn_hint = 0.f;        // a hint only helps below the radius
    if (cap < 1 || cap > D3F_NEIGHBOR_CAP) return D3F_ERR_ARG;
    if (!grid || !status_dev || !q_lens_dev || (Nq > 0 && (!queries || (width > 0 && !out)))) return D3F_ERR_ARG;
    if (flags & 1) { int rc0 = d3f_fill_u32(status_dev, 2, 0u, stream); if (rc0 != D3F_OK) return rc0; }
    const int want_kmax = (flags & D3F_NB_NO_KMAX) ? 0 : 1;
    if (Nq == 0) return D3F_OK;
    NbGrid g = nb_carve((void*)grid, grid_bytes, Ns, B);
    if (!g.ok) return D3F_ERR_WORKSPACE;
    const float4* qsorted = nullptr;
    bool same = false;
    if (query_grid == grid) {              // the queries ARE the supports
        if (Nq != Ns) return D3F_ERR_ARG;
        qsorted = g.sorted;
        same = true;
    } else if (query_grid) {
        NbGrid qg = nb_carve((void*)query_grid, query_grid_bytes, Nq, B);
        if (!qg.ok) return D3F_ERR_WORKSPACE;
        qsorted = qg.sorted;
    }
    return nb_search_dispatch(g, queries, Nq, q_lens_dev, B, radius, qsorted, same, out, ld, width, pad_value, cap, first_only, nn_hint,
                              want_kmax, status_dev, stream, (flags & D3F_NB_INTERNAL) != 0);
}

extern "C" int d3f_neighbor_grid_search(const void* grid, size_t grid_bytes, int Ns, const float* queries, int Nq,
                                        const int* q_lens_dev, int B, float radius, int queries_are_supports,
                                        int* out, int ld, int width, int pad_value, int cap, int first_only,
                                        float nn_hint, int reset_status, int* status_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Nq < 0 || Ns < 0 || B < 1 || B > D3F_MAX_BATCH || width < 0 || ld < width || !(radius >= 0.f)) return D3F_ERR_ARG;
    if (!(nn_hint >= 0.f) || nn_hint >= radius) nn_hint = 0.f;        // a hint only helps below the radius
    if (cap < 1 || cap > D3F_NEIGHBOR_CAP) return D3F_ERR_ARG;
    if (!grid || !status_dev || !q_lens_dev || (Nq > 0 && (!queries || (width > 0 && !out)))) return D3F_ERR_ARG;
    if (queries_are_supports && Nq != Ns) return D3F_ERR_ARG;
    if (reset_status & 1) { int rc0 = d3f_fill_u32(status_dev, 2, 0u, stream); if (rc0 != D3F_OK) return rc0; }
    const int want_kmax = (reset_status & D3F_NB_NO_KMAX) ? 0 : 1;
    if (Nq == 0) return D3F_OK;
    NbGrid g = nb_carve((void*)grid, grid_bytes, Ns, B);
    if (!g.ok) return D3F_ERR_WORKSPACE;
    return nb_search_dispatch(g, queries, Nq, q_lens_dev, B, radius, queries_are_supports ? g.sorted : nullptr, queries_are_supports != 0,
                              out, ld, width, pad_value, cap, first_only, nn_hint, want_kmax, status_dev, stream);
}

// ---- scoring of rigid-transform hypotheses against a built grid (downstream matching, registration.hip) --------------------
// For hypothesis v and source point i: p = R_v s_i + t_v; its nearest support strictly inside `radius` (same fp32 metric as the
// searches) is an inlier correspondence: count[v] += 1, sumd2[v] += d2 (fixed point, 2^-32 units: integer atomics make the
// sum independent of the order of arrival).  One thread per (v, i); the grid must hold ONE cloud (B = 1).
__global__ void __launch_bounds__(256) nb_score_kernel(const NbElem* __restrict__ el, const int* __restrict__ cell_start,
                                                       const int* __restrict__ cell_base, const float4* __restrict__ sorted,
                                                       const float* __restrict__ src, int Ns, const float* __restrict__ T, int V,
                                                       float r2, int* __restrict__ count, unsigned long long* __restrict__ sumd2,
                                                       int* __restrict__ nearest /* [Ns] or null, only meaningful for V == 1 */) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)V * Ns) return;
    const int v = (int)(tid / Ns), i = (int)(tid % Ns);
    const float* M = T + (size_t)v * 12;
    const float sx = src[3 * (size_t)i], sy = src[3 * (size_t)i + 1], sz = src[3 * (size_t)i + 2];
    const float qx = fmaf(M[0], sx, fmaf(M[1], sy, fmaf(M[2], sz, M[3])));
    const float qy = fmaf(M[4], sx, fmaf(M[5], sy, fmaf(M[6], sz, M[7])));
    const float qz = fmaf(M[8], sx, fmaf(M[9], sy, fmaf(M[10], sz, M[11])));
    const NbElem e = el[0];
    int cx, cy, cz;
    nb_cell_of(e, qx, qy, qz, cx, cy, cz);
    cx = min(max(cx, -2), e.dims[0] + 1);
    cy = min(max(cy, -2), e.dims[1] + 1);
    cz = min(max(cz, -2), e.dims[2] + 1);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, e.dims[0] - 1);
    float bd2 = 3.4e38f;
    int bidx = -1;
    if (x0 <= x1) {
        for (int j = 0; j < 9; ++j) {
            const int y = cy + (j % 3) - 1, z = cz + (j / 3) - 1;
            if (y < 0 || y >= e.dims[1] || z < 0 || z >= e.dims[2]) continue;
            const int rowbase = e.cbase + e.dims[0] * (y + e.dims[1] * z);
            const int lo = d3f_scan_at(cell_start, cell_base, rowbase + x0), hi = d3f_scan_at(cell_start, cell_base, rowbase + x1 + 1);
            for (int t = lo; t < hi; ++t) {
                const float4 sp = sorted[t];
                const float dx = __fsub_rn(qx, sp.x), dy = __fsub_rn(qy, sp.y), dz = __fsub_rn(qz, sp.z);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const int si = __float_as_int(sp.w);
                if (d2 < r2 && (d2 < bd2 || (d2 == bd2 && si < bidx))) { bd2 = d2; bidx = si; }
            }
        }
    }
    if (nearest && v == 0) nearest[i] = bidx;
    if (bidx >= 0) {
        atomicAdd(&count[v], 1);
        atomicAdd(&sumd2[v], (unsigned long long)((double)bd2 * 4294967296.0));
    }
}

// grid: built over the TARGET points (one cloud).  src f32[Ns,3]; T f32[V,12] row-major [R | t] per hypothesis.
// count_dev i32[V] / sumd2_dev u64[V] (2^-32 units) are OVERWRITTEN; nearest_dev i32[Ns] (optional): the correspondence of
// every source point under hypothesis 0 (-1: none inside the radius).
extern "C" int d3f_neighbor_grid_score(const void* grid, size_t grid_bytes, int Nt, const float* src, int Ns, const float* T,
                                       int V, float radius, int* count_dev, uint64_t* sumd2_dev, int* nearest_dev,
                                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Nt < 0 || Ns < 0 || V < 0 || !(radius >= 0.f)) return D3F_ERR_ARG;
    if (V == 0) return D3F_OK;
    if (!grid || !T || !count_dev || !sumd2_dev || (Ns > 0 && !src)) return D3F_ERR_ARG;
    NbGrid g = nb_carve((void*)grid, grid_bytes, Nt, 1);
    if (!g.ok) return D3F_ERR_WORKSPACE;
    int rc = d3f_fill_u32(count_dev, (size_t)V, 0u, stream);
    if (rc != D3F_OK) return rc;
    if ((rc = d3f_fill_u32(sumd2_dev, (size_t)V * 2, 0u, stream)) != D3F_OK) return rc;
    if (Ns == 0) return D3F_OK;
    nb_score_kernel<<<d3f_cdiv((long long)V * Ns, 256), 256, 0, stream>>>(g.el, g.cell_start, g.stmp, g.sorted, src, Ns, T, V,
                                                                         radius * radius, count_dev, (unsigned long long*)sumd2_dev, nearest_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" size_t d3f_radius_neighbors_workspace_bytes(int Nq, int Ns, int B) {
    (void)Nq;
    if (Ns < 0 || B < 1) return 0;
    return d3f_neighbor_grid_bytes(Ns, B) + 256;
}

// One-shot form (build + search), the direct replacement of the BatchOrderedNeighbors op.
extern "C" int d3f_batch_radius_neighbors(const float* queries, int Nq, const float* supports, int Ns,
                                          const int* q_lens_dev, const int* s_lens_dev, int B, float radius,
                                          int* out, int ld, int width, int pad_value, int* status_dev,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    if (Nq < 0 || Ns < 0 || B < 1 || B > D3F_MAX_BATCH || width < 0 || ld < width || !(radius >= 0.f)) return D3F_ERR_ARG;
    if (!status_dev || !q_lens_dev || !s_lens_dev || (Nq > 0 && (!queries || (width > 0 && !out))) || (Ns > 0 && !supports))
        return D3F_ERR_ARG;
    const size_t gb = d3f_neighbor_grid_bytes(Ns, B);
    if (!workspace || workspace_bytes < gb) return D3F_ERR_WORKSPACE;
    int rc = d3f_neighbor_grid_build(supports, Ns, s_lens_dev, B, radius, workspace, gb, stream_);
    if (rc != D3F_OK) return rc;
    return d3f_neighbor_grid_search(workspace, gb, Ns, queries, Nq, q_lens_dev, B, radius, 0, out, ld, width, pad_value,
                                    D3F_NEIGHBOR_CAP, 0, 0.f, 1, status_dev, stream_);
}
