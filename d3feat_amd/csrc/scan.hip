// Small shared primitives: exclusive scan (i32), per-element bounding boxes, length -> offset prefix.
// All HBM-bound streaming kernels; 256-thread blocks, 16 B per lane where the layout allows.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Exclusive scan, three launches: per-block scan (1024 items/block) -> scan of block sums (one block,
// chunked, carries a running total so any n works) -> add block offsets.
// ------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 4
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds /* >= 4 ints */, int* total) {
    // wave inclusive scan via DPP-free shuffles (wave = 64)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) lds[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
        int s = lds[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles_kernel(const int* in, int* out,
                                                                int n, int* __restrict__ block_sums) {
    __shared__ int lds[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan_256(s, lds, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_sums_kernel(int* __restrict__ block_sums, int nblocks,
                                                               int* __restrict__ total) {
    __shared__ int lds[4];
    int carry = 0;
    for (int c = 0; c < nblocks; c += SCAN_BLOCK) {
        int i = c + threadIdx.x;
        int v = (i < nblocks) ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan_256(v, lds, &tot);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_add_kernel(int* __restrict__ out, int n,
                                                              const int* __restrict__ block_sums) {
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    const int add = block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += add;
}

size_t d3f_scan_tmp_ints(int n) { return (size_t)d3f_cdiv(n > 0 ? n : 1, SCAN_TILE) + 64; }

int d3f_exclusive_scan_i32(const int* in, int* out, int n, int* tmp, int* total, hipStream_t stream) {
    if (n <= 0) {
        if (total) return d3f_fill_u32(total, 1, 0u, stream);
        return D3F_OK;
    }
    const int nb = d3f_cdiv(n, SCAN_TILE);
    scan_tiles_kernel<<<nb, SCAN_BLOCK, 0, stream>>>(in, out, n, tmp);
    scan_sums_kernel<<<1, SCAN_BLOCK, 0, stream>>>(tmp, nb, total);
    if (nb > 1) scan_add_kernel<<<nb, SCAN_BLOCK, 0, stream>>>(out, n, tmp);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// Fill / copy of 32-bit words as plain kernels.  Used instead of hipMemsetAsync / hipMemcpyAsync so that a captured launch
// sequence consists of kernel nodes only (no runtime-implemented memset / memcpy nodes inside the HIP graph).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fill_u32_kernel(unsigned* __restrict__ p, size_t n, unsigned v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = (((uintptr_t)p & 15) == 0) ? n / 4 : 0;
    uint4* p4 = (uint4*)p;
    const uint4 v4 = make_uint4(v, v, v, v);
    for (size_t j = i; j < n4; j += stride) p4[j] = v4;
    for (size_t j = n4 * 4 + i; j < n; j += stride) p[j] = v;
}

int d3f_fill_u32(void* p, size_t n_words, unsigned v, hipStream_t stream) {
    if (n_words == 0) return D3F_OK;
    long long blocks = d3f_cdiv((long long)(n_words / 4 + 1), 256);
    if (blocks > 2048) blocks = 2048;
    fill_u32_kernel<<<(int)blocks, 256, 0, stream>>>((unsigned*)p, n_words, v);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

__global__ void copy_i32_kernel(int* __restrict__ dst, const int* __restrict__ src, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int d3f_copy_i32(int* dst, const int* src, int n, hipStream_t stream) {
    if (n <= 0) return D3F_OK;
    copy_i32_kernel<<<d3f_cdiv(n, 256) > 64 ? 64 : d3f_cdiv(n, 256), 256, 0, stream>>>(dst, src, n);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// offs[0..B] from lens[0..B)   (B <= 255: one thread is plenty)
// ------------------------------------------------------------------------------------------------
__global__ void offsets_kernel(const int* __restrict__ lens, int B, int* __restrict__ offs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < B; ++b) { offs[b] = s; s += lens[b]; }
        offs[B] = s;
    }
}

int d3f_offsets_launch(const int* lens, int B, int* offs, hipStream_t stream) {
    offsets_kernel<<<1, 64, 0, stream>>>(lens, B, offs);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// Bounding boxes per batch element: grid (chunks, B).  Min/max are exact and order independent, so the
// result equals cpp_utils/cloud/cloud.cpp:27-66 (min_point / max_point) bit for bit.
// bbox must be pre-initialised: min slots 0xFFFFFFFF, max slots 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bbox_kernel(const float* __restrict__ pts, const int* __restrict__ offs,
                                                   unsigned* __restrict__ bbox) {
    const int b = blockIdx.y;
    const int lo = offs[b], hi = offs[b + 1];
    unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned u = d3f_f2ord(pts[3 * (size_t)i + d]);
            mn[d] = min(mn[d], u);
            mx[d] = max(mx[d], u);
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
}

__global__ void bbox_init_kernel(unsigned* __restrict__ bbox, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * 6) bbox[i] = ((i % 6) < 3) ? 0xFFFFFFFFu : 0u;
}

int d3f_bbox_launch(const float* pts, const int* offs, int B, int N, unsigned* bbox, hipStream_t stream) {
    bbox_init_kernel<<<d3f_cdiv(B * 6, 256), 256, 0, stream>>>(bbox, B);
    int chunks = d3f_cdiv(N > 0 ? N : 1, 256 * 4);
    if (chunks > 128) chunks = 128;
    bbox_kernel<<<dim3(chunks, B), 256, 0, stream>>>(pts, offs, bbox);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ---- trace marker ---------------------------------------------------------------------------------------------------
// A one-thread kernel whose only purpose is to show up BY NAME in a `rocprofv3 --kernel-trace`: bench.py launches it at the
// two ends of its timed region, tools/rocpd_summary.py --timed-region keeps the launches between the first two of them (the
// capture warm-ups, the calibration prologue and the untimed legs of a run stay out of the per-kernel table).
// One launch at the end of a replay: the device-resident sizes and status words of the whole launch sequence packed into ONE
// block (read back by one copy), and the sticky FLAG words (clear[clear_first + i * clear_step]; the size words beside them stay
// readable after the replay) cleared for the next replay -- instead of four copy nodes and a fill
// node per replay (r04: runtime copies and torch fills cost more kernel time than max pooling).
__global__ void __launch_bounds__(256) d3f_pack_status_kernel(int* __restrict__ dst, const int* __restrict__ a, int na,
                                                              const int* __restrict__ b, int nb, const int* __restrict__ c, int nc,
                                                              const int* __restrict__ d, int nd, int* __restrict__ clear, int nclear,
                                                              int clear_first, int clear_step) {
    for (int i = threadIdx.x; i < na; i += blockDim.x) dst[i] = a[i];
    for (int i = threadIdx.x; i < nb; i += blockDim.x) dst[na + i] = b[i];
    for (int i = threadIdx.x; i < nc; i += blockDim.x) dst[na + nb + i] = c[i];
    for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[na + nb + nc + i] = d[i];
    __syncthreads();                    // (`clear` may be one of the sources)
    for (int i = threadIdx.x; i < nclear; i += blockDim.x) clear[clear_first + i * clear_step] = 0;
}
extern "C" int d3f_pack_status(int* dst, const int* a, int na, const int* b, int nb, const int* c, int nc, const int* d, int nd,
                               int* clear, int nclear, int clear_first, int clear_step, void* stream) {
    if (!dst || na < 0 || nb < 0 || nc < 0 || nd < 0 || nclear < 0 || (na && !a) || (nb && !b) || (nc && !c) || (nd && !d) ||
        (nclear && (!clear || clear_first < 0 || clear_step < 1)))
        return D3F_ERR_ARG;
    d3f_pack_status_kernel<<<1, 256, 0, (hipStream_t)stream>>>(dst, a, na, b, nb, c, nc, d, nd, clear, nclear, clear_first, clear_step);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

__global__ void d3f_trace_marker_kernel(int id, int* sink) {
    if (sink && id == 0x7fffffff) *sink = id;
}
extern "C" int d3f_trace_marker(int id, void* stream) {
    d3f_trace_marker_kernel<<<1, 1, 0, (hipStream_t)stream>>>(id, nullptr);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
