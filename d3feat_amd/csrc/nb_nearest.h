// Nearest support inside the radius (column 0 of a neighbour row), round 6: FOUR LANES PER QUERY, one stencil row per lane.
// (included by radius_neighbors.hip; the same metric as every search: d2 = (dx*dx + dy*dy) + dz*dz in fp32 without FMA, strict
// d2 < r2, ties by the smaller support index -- column 0 of neighbors.cpp:125-208, all that closest_pool reads of the upsampling
// matrices, models/network_blocks.py:81.)
//
// Why (profiles/r06_experiments.txt n8-n9): the 16-lanes-per-query form spends ~126 vector instructions per query (run bounds
// shuffled across the group, the lane -> run select chain per candidate slot, group-wide ballots and a shuffle reduction) to look at
// ~35 candidates, and the vector pipe is what bounds it.  The cells the hint ball touches are at most 2 x 2 (y, z) rows of at most
// 2 cells when the hint is below half a cell -- the upsampling searches -- and each row is ONE contiguous run of the cell-sorted
// supports: lane j of the query's quad walks row j by itself (eight 16-byte records in flight), then the quad's four running
// minima are combined by two DPP steps.  ~15 vector instructions per query, the same four dependent round trips.  (One lane per
// query was measured too: 9 instructions per query, but its walk is a chain of ~10 round trips and even the largest launch of the
// pyramid has fewer than four wavefronts per SIMD to hide them -- 46 us against 59, and 4x slower on the small levels.)
// The queries come in the cell order of THEIR OWN grid when the caller has one (the level's conv grid), so neighbouring quads walk
// the same runs.  The result never depends on the hint: a query whose nearest support is not within it walks the full 27-cell
// stencil (rows j, j + 4, j + 8 per lane; rare).
#pragma once

struct NbNear {
    float bd2;
    int bidx;
    __device__ __forceinline__ void test(const float4 sp, float qx, float qy, float qz, float r2, bool valid) {
        const float dx = __fsub_rn(qx, sp.x), dy = __fsub_rn(qy, sp.y), dz = __fsub_rn(qz, sp.z);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const int si = __float_as_int(sp.w);
        if (valid && d2 < r2 && (d2 < bd2 || (d2 == bd2 && si < bidx))) { bd2 = d2; bidx = si; }
    }
    // supports [lo, hi) of the cell-sorted array, eight records in flight
    __device__ __forceinline__ void run(const float4* __restrict__ sorted, int lo, int hi, float qx, float qy, float qz, float r2) {
        for (int t = lo; t < hi; t += 8) {
            float4 c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = sorted[min(t + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) test(c[u], qx, qy, qz, r2, t + u < hi);
        }
    }
    // minimum over the four lanes of a quad (every lane ends with it)
    __device__ __forceinline__ void quad_min() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ctrl = s == 0 ? 0xB1 : 0x4E;       // lane ^ 1, lane ^ 2
            const float od = s == 0 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bd2), 0xB1, 0xF, 0xF, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bd2), 0x4E, 0xF, 0xF, true));
            const int oi = s == 0 ? __builtin_amdgcn_mov_dpp(bidx, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(bidx, 0x4E, 0xF, 0xF, true);
            (void)ctrl;
            if (od < bd2 || (od == bd2 && oi < bidx)) { bd2 = od; bidx = oi; }
        }
    }
};

template <bool HINT>
__global__ void __launch_bounds__(256)
nb_nearest_kernel(const float* __restrict__ q, int Nq, const int* __restrict__ qlens, int B, const NbElem* __restrict__ el,
                  const int* __restrict__ cell_start, const int* __restrict__ cell_base, const float4* __restrict__ sorted,
                  const float4* __restrict__ qsorted, float r2, int pad, const int* __restrict__ ns_dev, int* __restrict__ out, int ld,
                  int width, float nn_hint, const int* __restrict__ inv) {
    // inv != NULL: INTERNAL numbering -- row p (the visit position) instead of row qi, the entry = inv[index]
    __shared__ NbElem sel[NB_EL_LDS];
    __shared__ int sOff[NB_EL_LDS + 1];
    const bool staged = B <= NB_EL_LDS;
    if (staged) {
        constexpr int W = (int)(sizeof(NbElem) / sizeof(int));
        for (int i = threadIdx.x; i < B * W; i += blockDim.x) ((int*)sel)[i] = ((const int*)el)[i];
        if (threadIdx.x <= B) {
            int s = 0;
            for (int j = 0; j < (int)threadIdx.x; ++j) s += qlens[j];
            sOff[threadIdx.x] = s;
        }
        __syncthreads();
    }
    int nq_real = 0;
    for (int j = 0; j < B; ++j) nq_real += qlens[j];
    const int nq = min(Nq, nq_real);                      // Nq is the capacity, sum(qlens) the real number of queries
    const int nblk = (nq + 63) >> 6;                     // 64 queries per workgroup, four lanes each
    if ((int)blockIdx.x >= nblk) return;
    // one contiguous run of query blocks per XCD (common.h): neighbouring blocks walk the same runs
    const int p = (int)d3f_xcd_tile(blockIdx.x, (unsigned)nblk) * 64 + (int)(threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    if (p >= nq) return;                                  // (whole quads leave together)
    if (pad == D3F_PAD_NUM_SUPPORTS) pad = *ns_dev;
    float qx, qy, qz;
    int qi = p;
    if (qsorted) {            // the p-th query in the cell order of the queries' own grid: position and index in one record
        const float4 me = qsorted[p];
        qx = me.x; qy = me.y; qz = me.z; qi = __float_as_int(me.w);
    } else {
        qx = q[3 * (size_t)p]; qy = q[3 * (size_t)p + 1]; qz = q[3 * (size_t)p + 2];
    }
    int b = 0;                // batch element: the last one starting at or before the query's index
    if (staged) {
        for (int j = 1; j < B; ++j) b = (qi >= sOff[j]) ? j : b;
    } else {
        for (int j = 1, start = qlens[0]; j < B; ++j) { if (qi >= start) b = j; start += qlens[j]; }
    }
    const NbElem e = staged ? sel[b] : el[b];
    int cx, cy, cz;
    nb_cell_of(e, qx, qy, qz, cx, cy, cz);
    cx = min(max(cx, -2), e.dims[0] + 1);
    cy = min(max(cy, -2), e.dims[1] + 1);
    cz = min(max(cz, -2), e.dims[2] + 1);
    NbNear best;
    best.bd2 = 3.4e38f;
    best.bidx = 0x7fffffff;
    bool found = false;
    if (HINT) {
        // cells touched by the ball of radius nn_hint, in the grid's own fp64 index arithmetic (monotone, so a support within
        // nn_hint of the query lies in [lo, hi] on every axis): at most two per axis when nn_hint < cell / 2
        const double h = (double)nn_hint;
        const int xl = max(max(cx - 1, (int)floor(((double)qx - h - e.mn[0]) * e.inv_h)), 0);
        const int xh = min(min(cx + 1, (int)floor(((double)qx + h - e.mn[0]) * e.inv_h)), e.dims[0] - 1);
        const int yl = max(max(cy - 1, (int)floor(((double)qy - h - e.mn[1]) * e.inv_h)), 0);
        const int yh = min(min(cy + 1, (int)floor(((double)qy + h - e.mn[1]) * e.inv_h)), e.dims[1] - 1);
        const int zl = max(max(cz - 1, (int)floor(((double)qz - h - e.mn[2]) * e.inv_h)), 0);
        const int zh = min(min(cz + 1, (int)floor(((double)qz + h - e.mn[2]) * e.inv_h)), e.dims[2] - 1);
        if (xl <= xh && yl <= yh && zl <= zh) {
            // (yh - yl + 1) x (zh - zl + 1) <= 3 x 3 rows (2 x 2 for a hint below half a cell): lane `part` walks rows part, part + 4,
            // part + 8; the bounds of a lane's rows are fetched together
            const int ny = yh - yl + 1, nrows = ny * (zh - zl + 1);
            int lo[3], hi[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int j = part + 4 * u;
                lo[u] = hi[u] = 0;
                if (j < nrows) {
                    const int y = yl + j % ny, z = zl + j / ny;
                    const int rowbase = e.cbase + e.dims[0] * (y + e.dims[1] * z);
                    lo[u] = d3f_scan_at(cell_start, cell_base, rowbase + xl);
                    hi[u] = d3f_scan_at(cell_start, cell_base, rowbase + xh + 1);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) best.run(sorted, lo[u], hi[u], qx, qy, qz, r2);
        }
        best.quad_min();
        // the restricted walk stands when its nearest support lies within the hint (0.998: the fp32 d2 of a support just outside
        // the visited cells can round below that of one just inside the ball); else: the full stencil
        found = best.bidx != 0x7fffffff && best.bd2 <= 0.998f * nn_hint * nn_hint;
    }
    if (!found) {
        best.bd2 = 3.4e38f;
        best.bidx = 0x7fffffff;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, e.dims[0] - 1);
        if (x0 <= x1) {
            for (int j = part; j < 9; j += 4) {
                const int y = cy + (j % 3) - 1, z = cz + (j / 3) - 1;
                if (y < 0 || y >= e.dims[1] || z < 0 || z >= e.dims[2]) continue;
                const int rowbase = e.cbase + e.dims[0] * (y + e.dims[1] * z);
                const int lo = d3f_scan_at(cell_start, cell_base, rowbase + x0), hi = d3f_scan_at(cell_start, cell_base, rowbase + x1 + 1);
                best.run(sorted, lo, hi, qx, qy, qz, r2);
            }
        }
        best.quad_min();
        found = best.bidx != 0x7fffffff;
    }
    int* row = out + (size_t)(inv ? p : qi) * ld;
    if (part == 0 && width > 0) row[0] = found ? (inv ? inv[best.bidx] : best.bidx) : pad;
    for (int j = 1 + part; j < width; j += 4) row[j] = pad;
}
