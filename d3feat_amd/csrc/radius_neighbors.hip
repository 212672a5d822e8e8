// Fixed-radius neighbour search on gfx950: uniform cell grid + one wavefront per query.
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
// MI355X design (HBM/L2-bound gather work):
//   build:  supports -> cell id (fp64 index arithmetic, cell edge slightly > radius so the 27-cell stencil is
//           a guaranteed superset) -> counting sort by cell (atomic histogram, exclusive scan, scatter) into a
//           float4 {x,y,z,index-bits} array: one 16-byte load per candidate, x-adjacent cells are contiguous,
//           so the 27-cell stencil is 9 contiguous runs;
//   search: one 64-lane wavefront per query streams the 9 runs, 64 candidates per step; hits are compacted
//           into the wave's LDS segment with a ballot + popcount prefix (variable-length lists); the <=K hits
//           are ordered by rank counting (each lane counts how many hits precede its own -- keys are unique so
//           ranks are a permutation) and the first `width` ranks are stored straight into the output row.
#include "common.h"

struct NbElem {
    double mn[3];
    double inv_h;
    int dims[3];
    int cbase;  // first cell of this element in the global cell array
};

#define NB_WAVES_PER_BLOCK 4

// one thread per element: choose the cell edge (>= radius*(1+2^-20); doubled until the element's grid fits
// its share of the cell budget), grid dims and cell base.
__global__ void nb_prep_kernel(const unsigned* __restrict__ bbox, const int* __restrict__ soffs, int B, float radius,
                               long long cell_budget, NbElem* __restrict__ el, int* __restrict__ ncells_total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
                e.mn[d] = (double)d3f_ord2f(bbox[b * 6 + d]);
                mx[d] = (double)d3f_ord2f(bbox[b * 6 + 3 + d]);
            }
            double h = (double)radius * (1.0 + 1.0 / 1048576.0);
            if (!(h > 0.0)) h = 1.0;
            for (int it = 0; it < 64; ++it) {
                // dims use the same expression as nb_cell_of, so every support's cell is in range by monotonicity
                e.inv_h = 1.0 / h;
                double tot = 1.0;
                for (int d = 0; d < 3; ++d) {
                    double n = floor((mx[d] - e.mn[d]) * e.inv_h) + 1.0;
                    e.dims[d] = (n < 1073741824.0) ? (int)n : 1073741824;
                    tot *= n;
                }
                if (tot <= (double)per) break;
                h *= 2.0;
            }
        }
        e.cbase = (int)base;
        base += (long long)e.dims[0] * e.dims[1] * e.dims[2];
        el[b] = e;
    }
    *ncells_total = (int)base;
}

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
    if (i >= Ns) return;
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

__global__ void __launch_bounds__(256) nb_scatter_kernel(const float* __restrict__ s, int Ns, const int* __restrict__ cell_of,
                                                         const int* __restrict__ cell_start, int* __restrict__ cell_cur,
                                                         float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ns) return;
    const int c = cell_of[i];
    const int pos = cell_start[c] + atomicAdd(&cell_cur[c], 1);
    sorted[pos] = make_float4(s[3 * (size_t)i], s[3 * (size_t)i + 1], s[3 * (size_t)i + 2], __int_as_float(i));
}

// ---- search: one wavefront per query -----------------------------------------------------------------
__global__ void __launch_bounds__(64 * NB_WAVES_PER_BLOCK)
nb_search_kernel(const float* __restrict__ q, int Nq, const int* __restrict__ qoffs, int B,
                 const NbElem* __restrict__ el, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
                 float r2, int pad, int* __restrict__ out, int ld, int width, int cap, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* hd2 = (float*)smem + (size_t)wave * 2 * cap;
    int* hidx = (int*)hd2 + cap;
    const int qi = blockIdx.x * NB_WAVES_PER_BLOCK + wave;
    if (qi >= Nq) return;
    const int b = d3f_find_elem(qoffs, B, qi);
    const NbElem e = el[b];
    const float qx = q[3 * (size_t)qi], qy = q[3 * (size_t)qi + 1], qz = q[3 * (size_t)qi + 2];
    int cx, cy, cz;
    nb_cell_of(e, qx, qy, qz, cx, cy, cz);
    cx = min(max(cx, -2), e.dims[0] + 1);
    cy = min(max(cy, -2), e.dims[1] + 1);
    cz = min(max(cz, -2), e.dims[2] + 1);
    int n = 0;  // wave-uniform hit count
    const unsigned long long lt = d3f_lanemask_lt();
    // a query outside the supports' box by more than one cell cannot have neighbours: the clamped loops are empty
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, e.dims[0] - 1);
    if (x0 <= x1) {
        for (int z = max(cz - 1, 0); z <= min(cz + 1, e.dims[2] - 1); ++z) {
            for (int y = max(cy - 1, 0); y <= min(cy + 1, e.dims[1] - 1); ++y) {
                const int rowbase = e.cbase + e.dims[0] * (y + e.dims[1] * z);
                const int lo = cell_start[rowbase + x0], hi = cell_start[rowbase + x1 + 1];
                for (int t0 = lo; t0 < hi; t0 += 64) {
                    const int t = t0 + lane;
                    bool hit = false;
                    float d2 = 0.f;
                    int si = 0;
                    if (t < hi) {
                        const float4 sp = sorted[t];
                        const float dx = __fsub_rn(qx, sp.x), dy = __fsub_rn(qy, sp.y), dz = __fsub_rn(qz, sp.z);
                        d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                        si = __float_as_int(sp.w);
                        hit = d2 < r2;
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const int pos = n + __popcll(m & lt);
                        if (pos < cap) { hd2[pos] = d2; hidx[pos] = si; }
                    }
                    n += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) {
        atomicMax(&status[0], n);
        if (n > cap) atomicOr(&status[1], D3F_ST_HIT_OVERFLOW);
    }
    const int m = min(n, cap);
    // the wave's own LDS writes are visible to its own later reads once lgkmcnt drains; no other wave touches them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int* row = out + (size_t)qi * ld;
    for (int e0 = 0; e0 < m; e0 += 64) {
        const int ei = e0 + lane;
        if (ei < m) {
            const float kd = hd2[ei];
            const int ki = hidx[ei];
            int rank = 0;
            for (int j = 0; j < m; ++j) {
                const float dj = hd2[j];
                const int ij = hidx[j];
                rank += (dj < kd || (dj == kd && ij < ki)) ? 1 : 0;
            }
            if (rank < width) row[rank] = ki;
        }
    }
    for (int j = m + lane; j < width; j += 64) row[j] = pad;
}

// ------------------------------------------------------------------------------------------------
static long long nb_cell_budget(int Ns) {
    long long b = 16ll * (long long)(Ns > 0 ? Ns : 1);
    if (b < (1ll << 16)) b = 1ll << 16;
    if (b > (1ll << 28)) b = 1ll << 28;
    return b;
}

extern "C" size_t d3f_radius_neighbors_workspace_bytes(int Nq, int Ns, int B) {
    (void)Nq;
    if (Ns < 0 || B < 1) return 0;
    const long long cells = nb_cell_budget(Ns) + 8;
    size_t ns = (size_t)(Ns > 0 ? Ns : 1);
    size_t bytes = 0;
    bytes += 2 * d3f_align((B + 1) * sizeof(int));
    bytes += d3f_align(B * 6 * sizeof(unsigned));
    bytes += d3f_align(B * sizeof(NbElem));
    bytes += d3f_align(64);
    bytes += 3 * d3f_align((size_t)cells * sizeof(int));
    bytes += d3f_align(ns * sizeof(int));
    bytes += d3f_align(ns * sizeof(float4));
    bytes += d3f_align(d3f_scan_tmp_ints((int)cells) * sizeof(int));
    return bytes + 4096;
}

extern "C" int d3f_batch_radius_neighbors(const float* queries, int Nq, const float* supports, int Ns,
                                          const int* q_lens_dev, const int* s_lens_dev, int B, float radius,
                                          int* out, int ld, int width, int pad_value, int* status_dev,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Nq < 0 || Ns < 0 || B < 1 || B > D3F_MAX_BATCH || width < 0 || ld < width || !(radius >= 0.f)) return D3F_ERR_ARG;
    if (!status_dev || !q_lens_dev || !s_lens_dev || (Nq > 0 && (!queries || (width > 0 && !out))) || (Ns > 0 && !supports))
        return D3F_ERR_ARG;
    D3F_HIP_TRY(hipMemsetAsync(status_dev, 0, 2 * sizeof(int), stream));
    if (Nq == 0) return D3F_OK;
    const long long budget = nb_cell_budget(Ns);
    const long long cells = budget + 8;
    D3fArena ar(workspace, workspace_bytes);
    const size_t ns = (size_t)(Ns > 0 ? Ns : 1);
    int* qoffs = ar.take<int>(B + 1);
    int* soffs = ar.take<int>(B + 1);
    unsigned* bbox = ar.take<unsigned>(B * 6);
    NbElem* el = ar.take<NbElem>(B);
    int* ncells = ar.take<int>(16);
    int* cell_cnt = ar.take<int>((size_t)cells);
    int* cell_start = ar.take<int>((size_t)cells);
    int* cell_cur = ar.take<int>((size_t)cells);
    int* cell_of = ar.take<int>(ns);
    float4* sorted = ar.take<float4>(ns);
    int* stmp = ar.take<int>(d3f_scan_tmp_ints((int)cells));
    if (!ar.ok) return D3F_ERR_WORKSPACE;

    int rc;
    if ((rc = d3f_offsets_launch(q_lens_dev, B, qoffs, stream)) != D3F_OK) return rc;
    if ((rc = d3f_offsets_launch(s_lens_dev, B, soffs, stream)) != D3F_OK) return rc;
    if ((rc = d3f_bbox_launch(supports, soffs, B, Ns, bbox, stream)) != D3F_OK) return rc;
    nb_prep_kernel<<<1, 64, 0, stream>>>(bbox, soffs, B, radius, budget, el, ncells);
    // The cell arrays are sized for the whole budget; scanning all of it keeps the launch shapes static
    // (no host read-back of the real cell count).  cell_start[c] for c >= ncells is the total count.
    D3F_HIP_TRY(hipMemsetAsync(cell_cnt, 0, (size_t)cells * sizeof(int), stream));
    D3F_HIP_TRY(hipMemsetAsync(cell_cur, 0, (size_t)cells * sizeof(int), stream));
    if (Ns > 0) {
        nb_count_kernel<<<d3f_cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, soffs, B, el, cell_of, cell_cnt);
        D3F_LAUNCH_CHECK();
    }
    if ((rc = d3f_exclusive_scan_i32(cell_cnt, cell_start, (int)cells, stmp, nullptr, stream)) != D3F_OK) return rc;
    if (Ns > 0) {
        nb_scatter_kernel<<<d3f_cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, cell_of, cell_start, cell_cur, sorted);
        D3F_LAUNCH_CHECK();
    }
    const int cap = D3F_NEIGHBOR_CAP;
    const size_t lds = (size_t)NB_WAVES_PER_BLOCK * cap * 2 * sizeof(float);
    const float r2 = radius * radius;
    nb_search_kernel<<<d3f_cdiv(Nq, NB_WAVES_PER_BLOCK), 64 * NB_WAVES_PER_BLOCK, lds, stream>>>(
        queries, Nq, qoffs, B, el, cell_start, sorted, r2, pad_value, out, ld, width, cap, status_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
