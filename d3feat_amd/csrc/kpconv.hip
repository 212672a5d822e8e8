// KPConv on gfx950: neighbour gather + kernel-point influences + weighted aggregation (+ fused contraction / epilogue).
//
// Reference: kernels/convolution_ops.py:161-255 (KPConv_ops).  The TF graph materialises
// [n, K, 15, 3] differences and an [n, K, Cin] gather in memory (:200-205, :237); here both stay on chip:
//   wf[n,p,c]  = sum_k h(|| (s[idx[n,k]] - q[n]) - KP[p] ||) * f[idx[n,k], c]          (:194-240)
//   inv_cnt[n] = 1 / max(#{k : sum_c f[idx[n,k], c] > 0}, 1)                          (:250-252)
//   out[n,o]   = (sum_{p,c} wf[n,p,c] * K_values[p,c,o]) * inv_cnt[n]                 (:243-253)
//
// Kernels (VALU / gather-latency bound; the feature rows live in L2 / Infinity Cache after the first touch):
//  * kpconv_agg_vec4<LQ> (Cin % 4 == 0): a 256-thread workgroup owns TQ = 256/LQ queries, LQ = Cin/4 lanes per
//    query, each lane owns 4 channels x 15 kernel points (60 accumulators in VGPRs).  Neighbours are processed
//    in chunks of KC = LQ: first every thread computes the 15 influences of ONE (query, neighbour) pair and
//    parks them in LDS (64 B per pair) together with the neighbour index; then each lane walks its query's
//    chunk: coalesced 16-byte feature loads (the LQ lanes of a query read one contiguous Cin*4-byte row), eight
//    requested before any is consumed, 4 LDS b128 broadcasts for the influences, 60 FMAs.  Shadow neighbours
//    (idx >= Ns) are skipped: their influence is exactly 0 in the reference (shadow point at 1e6), their row is 0.
//    Writes wf / inv_cnt; the contraction runs on the matrix cores in gemm_f32.hip (row_scale = inv_cnt).
//  * kpconv_fused32_kernel (Cin = Cout = 32, the level-0 convolutions): the same phases, then the 32 x 480 wf tile is
//    contracted from LDS with v_mfma_f32_32x32x2_f32 and the BN / LeakyReLU epilogue applied: wf never reaches HBM.
//  * kpconv_c1_kp_kernel (Cin = 1, the input layer, aggregation 'sum'): lanes = (query, kernel point), no cross-lane
//    reduction; kpconv_c1_fused_kernel (lanes = neighbours) serves aggregation 'closest'.
//  * kpconv_agg_scalar: any other Cin, one thread per (query, channel).
// Queries are visited in the cell-sorted order of the neighbour grid when the caller passes it (q_order).
#include "common.h"
#include <cstdlib>

#define KP_MAXP D3F_NUM_KP_MAX  // 16 slots, 15 used by the reference

struct KpParams {
    float kp[KP_MAXP * 3];
    float kx[KP_MAXP], ky[KP_MAXP], kz[KP_MAXP];   // the same points, one array per coordinate: (kx[p], kx[p+1]) is a register pair
    int num_kp;
    float extent;
    float inv_2extent;  // 1 / (2 * extent)
    int influence;    // 0 constant, 1 linear, 2 gaussian
    int aggregation;  // 0 sum, 1 closest
};

// influences of one neighbour (relative position r) for all kernel points
__device__ __forceinline__ void kp_influences(const KpParams& P, float rx, float ry, float rz, float* w) {
    float best = 3.4e38f;
    int bestp = 0;
#pragma unroll
    for (int p = 0; p < KP_MAXP; ++p) {
        if (p < P.num_kp) {
            const float dx = rx - P.kp[3 * p], dy = ry - P.kp[3 * p + 1], dz = rz - P.kp[3 * p + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            float v;
            // v_sqrt_f32 (1 ulp) and a reciprocal multiply instead of the correctly rounded sqrt / divide sequences:
            // ~1e-7 relative on a weight in [0,1], four orders below the 1e-4 parity bar, and 3x fewer VALU instructions
            if (P.influence == 1) v = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2 + 1e-10f) * P.inv_2extent, 0.0f);
            else if (P.influence == 0) v = 1.0f;
            else { const float sig = P.extent * 0.3f; v = expf(-d2 / (2.0f * sig * sig + 1e-9f)); }
            w[p] = v;
            if (d2 < best) { best = d2; bestp = p; }
        } else {
            w[p] = 0.f;
        }
    }
    if (P.aggregation == 1) {
#pragma unroll
        for (int p = 0; p < KP_MAXP; ++p)
            if (p != bestp) w[p] = 0.f;
    }
}

// The configuration every shipped model uses (linear influence, 'sum' aggregation, 15 kernel points -- parameters.txt:
// KP_influence = linear, convolution_mode = sum, num_kernel_points = 15) as straight-line code: 15 x {3 subtractions,
// 3 multiply-adds, add, v_sqrt, multiply-subtract, max}, no mode branches, the kernel points read as scalar operands.
// FAST = false keeps the general function above (other modes / fewer kernel points).
// These kernels are bound by VALU issue (r03 x14: 597 vector instructions per 8-neighbour chunk and wavefront, 285 of them the
// aggregation's FMAs), so the influences are computed two kernel points at a time with the packed fp32 forms (v_pk_add / v_pk_mul /
// v_pk_fma_f32: two IEEE operations per lane and issue slot, the same roundings as the scalar sequence -> bit-identical results).
typedef float kp_f2 __attribute__((ext_vector_type(2)));
template <bool FAST>
__device__ __forceinline__ void kp_influences_t(const KpParams& P, float rx, float ry, float rz, float* w) {
    if (FAST) {
        const kp_f2 rx2 = {rx, rx}, ry2 = {ry, ry}, rz2 = {rz, rz};
        const kp_f2 one = {1.0f, 1.0f}, ninv = {-P.inv_2extent, -P.inv_2extent}, tiny = {1e-10f, 1e-10f};
#pragma unroll
        for (int p = 0; p < KP_MAXP - 2; p += 2) {
            const kp_f2 kx = {P.kx[p], P.kx[p + 1]}, ky = {P.ky[p], P.ky[p + 1]}, kz = {P.kz[p], P.kz[p + 1]};
            const kp_f2 dx = rx2 - kx, dy = ry2 - ky, dz = rz2 - kz;
            const kp_f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)) + tiny;
            const kp_f2 sq = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
            const kp_f2 v = __builtin_elementwise_fma(sq, ninv, one);      // fma(-sqrt, 1/(2 extent), 1) as the scalar form
            w[p] = fmaxf(v.x, 0.0f);
            w[p + 1] = fmaxf(v.y, 0.0f);
        }
        {
            constexpr int p = KP_MAXP - 2;
            const float dx = rx - P.kx[p], dy = ry - P.ky[p], dz = rz - P.kz[p];
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            w[p] = fmaxf(fmaf(-__builtin_amdgcn_sqrtf(d2 + 1e-10f), P.inv_2extent, 1.0f), 0.0f);
        }
        w[KP_MAXP - 1] = 0.f;
    } else {
        kp_influences(P, rx, ry, rz, w);
    }
}
static inline KpParams kp_make_params(const float* kp_host, int num_kp, float KP_extent, int influence, int aggregation) {
    KpParams P;
    for (int i = 0; i < KP_MAXP * 3; ++i) P.kp[i] = i < num_kp * 3 ? kp_host[i] : 0.f;
    for (int p = 0; p < KP_MAXP; ++p) { P.kx[p] = P.kp[3 * p]; P.ky[p] = P.kp[3 * p + 1]; P.kz[p] = P.kp[3 * p + 2]; }
    P.num_kp = num_kp; P.extent = KP_extent; P.inv_2extent = 1.0f / (2.0f * KP_extent); P.influence = influence;
    P.aggregation = aggregation;
    return P;
}
// (row addressing by one 24-bit multiply: common.h, d3f_fits_u24)
// (and the feature matrix must fit a buffer resource: rows * ld * 4 bytes < 2^32)
static inline bool kp_fits_u24(int Nq, int Ns, int ld_idx, int ldf) {
    return d3f_fits_u24(Nq, ld_idx) && d3f_fits_u24(Ns, ldf) && (long long)Ns * ldf < (1ll << 30);
}
// one 16-byte piece (channels c4 .. c4+3) of feature row `id`, fetched with a BUFFER load: the feature matrix is described by a
// buffer resource (base + size in SGPRs), the lane supplies a 32-bit byte offset -- no 64-bit address arithmetic per gather -- and
// a shadow neighbour (id < 0) supplies an offset beyond the buffer: the hardware's range check returns exact zeros for it, with no
// branch, no select and no zero fill.  (Round 3 read row 0 for shadows and relied on their influences being exactly 0: a
// non-finite value in row 0 turned 0 * Inf into NaN for every query with a shadow slot -- ADVICE r03.)
template <class FT> struct KpFeatBuf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ KpFeatBuf(const FT* f, int rows, int ldf) {
        const unsigned long long bytes = (unsigned long long)(rows > 0 ? rows : 1) * (unsigned)ldf * sizeof(FT);   // < 2^32: kp_fits_u24
        r = __builtin_amdgcn_make_buffer_rsrc((void*)f, 0, (int)(unsigned)bytes, 0x00020000);
    }
};
__device__ __forceinline__ float4 kp_gather4(const KpFeatBuf<float>& B, int id, int ldf, int c4) {
    const unsigned off = id >= 0 ? (__umul24((unsigned)id, (unsigned)ldf) + (unsigned)c4) * 4u : 0xfffffff0u;
    typedef unsigned kp_u4 __attribute__((ext_vector_type(4)));
    const kp_u4 v = __builtin_amdgcn_raw_buffer_load_b128(B.r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 kp_gather4(const KpFeatBuf<unsigned short>& B, int id, int ldf, int c4) {
    const unsigned off = id >= 0 ? (__umul24((unsigned)id, (unsigned)ldf) + (unsigned)c4) * 2u : 0xfffffff0u;
    typedef unsigned kp_u2 __attribute__((ext_vector_type(2)));
    const kp_u2 w = __builtin_amdgcn_raw_buffer_load_b64(B.r, (int)off, 0, 0);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
}
static inline bool kp_fast_config(int num_kp, int influence, int aggregation) {
    return num_kp == KP_MAXP - 1 && influence == 1 && aggregation == 0;
}

// Phase-A operands of one (query, neighbour) pair, fetched AHEAD of their chunk so that no chunk starts with a chain of dependent
// round trips (index -> support point): the index of chunk j + 1 is requested at the top of chunk j (it depends on nothing), the
// point and row flag it names at the top of chunk j's phase B (the index has had phase A to arrive) and land under phase B's
// gathers and FMAs.  A shadow neighbour (index outside [0, Ns)) reads row 0 and is moved to 1e6 -- the reference's own shadow
// point (kernels/convolution_ops.py:186-188): every influence of the shipped (linear) configuration is exactly 0, no branch.
struct KpPair {
    int id;              // neighbour index, -1: shadow
    float x, y, z;       // support point
    bool pos;            // row flag of the support (the neighbour-count test)
};
__device__ __forceinline__ int kp_pair_index(const int* __restrict__ idrow, bool live, int k, int K, int Ns) {
    return (live && k < K) ? idrow[k] : Ns;
}
__device__ __forceinline__ KpPair kp_pair_fetch(int id, int Ns, const float* __restrict__ s, const unsigned char* __restrict__ rowpos) {
    KpPair r;
    const bool ok = id >= 0 && id < Ns;
    const unsigned row = ok ? (unsigned)id : 0u;
    const float* sp = s + 3u * row;
    r.x = sp[0]; r.y = sp[1]; r.z = sp[2];
    r.pos = rowpos[row] != 0;
    r.id = ok ? id : -1;
    return r;
}
template <bool FAST>
__device__ __forceinline__ bool kp_pair_influences(const KpParams& P, const KpPair& pr, float qx, float qy, float qz, float* w) {
    const bool ok = pr.id >= 0;
    kp_influences_t<FAST>(P, ok ? pr.x - qx : 1e6f, ok ? pr.y - qy : 1e6f, ok ? pr.z - qz : 1e6f, w);
    if (!FAST) {        // 'constant' influence is 1 at any distance, 'closest' picks a kernel point anyway: the general modes zero
#pragma unroll          // a shadow's weights explicitly (in the reference its FEATURE row is the zero row)
        for (int p = 0; p < KP_MAXP; ++p) w[p] = ok ? w[p] : 0.f;
    }
    return ok && pr.pos;
}

// The 16 influences of one (query, neighbour) pair in LDS: four 16-byte quads at base + 16*slot floats.  Eight adjacent
// lanes store their pairs with ds_write_b128 at a 64-byte stride, i.e. on two bank groups only (4-way conflict: a third of the
// LDS-active cycles of these kernels); rotating the quad order by slot/2 spreads them over all eight.  Readers undo it.
__device__ __forceinline__ void kp_store_w(float* __restrict__ pair_base, int slot, const float* w) {
    float4* dst = (float4*)pair_base;
    const int r = slot >> 1;
    dst[(0 + r) & 3] = make_float4(w[0], w[1], w[2], w[3]);
    dst[(1 + r) & 3] = make_float4(w[4], w[5], w[6], w[7]);
    dst[(2 + r) & 3] = make_float4(w[8], w[9], w[10], w[11]);
    dst[(3 + r) & 3] = make_float4(w[12], w[13], w[14], w[15]);
}
__device__ __forceinline__ void kp_load_w(const float* __restrict__ pair_base, int slot, float* w) {
    const float4* src = (const float4*)pair_base;
    const int r = slot >> 1;
    const float4 w0 = src[(0 + r) & 3], w1 = src[(1 + r) & 3], w2 = src[(2 + r) & 3], w3 = src[(3 + r) & 3];
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
    w[8] = w2.x; w[9] = w2.y; w[10] = w2.z; w[11] = w2.w; w[12] = w3.x; w[13] = w3.y; w[14] = w3.z; w[15] = w3.w;
}

// Neighbour count of a query (the reference's `sum_c f > 0` test, :250-252) in phase A, where LQ consecutive lanes hold LQ
// neighbours of one query: one wavefront ballot and a popcount by the query's first lane -- the sole writer of lcnt[ql]
// between two barriers -- instead of one LDS atomic per neighbour on a shared word.  Must be reached by every lane.
template <int LQ>
__device__ __forceinline__ void kp_count_positive(bool positive, int ql, int cl, int* __restrict__ lcnt) {
    const unsigned long long m = __ballot(positive);
    if (LQ >= 64) {
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&lcnt[ql], __popcll(m));      // a query spans LQ / 64 wavefronts
    } else if (cl == 0) {
        const int sh = (threadIdx.x & 63) & ~(LQ - 1);
        const int c = __popcll((m >> sh) & ((1ull << (LQ & 63)) - 1ull));
        if (c) lcnt[ql] += c;
    }
}

// row_pos[s] = (sum_c f[s,c] > 0) ? 1 : 0  -- the reference's neighbour-count test (:250-251) depends only on
// the support row, so it is evaluated once per support instead of once per (query, neighbour).
// The test is discontinuous: a row whose fp32 sum lies within rounding of 0 flips with the summation order, and the
// reference's own order (Eigen's reduction tree inside tf.reduce_sum) is an implementation detail.  The sum is therefore
// accumulated in fp64 -- exact for <= 2^20 fp32 terms of comparable exponent, so its sign is the sign of the real-number
// sum whatever the order -- which every fp32 order agrees with wherever that order's own result is not rounding noise.
__global__ void __launch_bounds__(256) kp_rowpos_kernel(const float* __restrict__ f, int Ns, const int* __restrict__ Ns_dev,
                                                        int ldf, int Cin, unsigned char* __restrict__ pos) {
    // one wavefront per row
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= d3f_dyn(Ns, Ns_dev)) return;
    double s = 0.0;
    for (int c = lane; c < Cin; c += 64) s += (double)f[(size_t)row * ldf + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) pos[row] = s > 0.0 ? 1 : 0;
}

// 16-byte loads, LPR lanes per row (64 / LPR rows per wavefront): for the narrow rows of the fine levels the one-row-per-
// wavefront form spends its time in the 64-lane fp64 butterfly, not in the loads.  Same fp64 sum, another (equally exact) order.
template <int LPR, class FT = float>   // FT: feature storage type (float, or unsigned short = bf16: common.h D3fFeat)
__global__ void __launch_bounds__(256) kp_rowpos_vec_kernel(const FT* __restrict__ f, int Ns, const int* __restrict__ Ns_dev,
                                                            int ldf, int Cin, unsigned char* __restrict__ pos) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = t / LPR;
    const int l = (int)(t % LPR);
    const bool in = row < (long long)d3f_dyn(Ns, Ns_dev);
    double s = 0.0;
    if (in) {
        for (int c = 4 * l; c < Cin; c += 4 * LPR) {
            const float4 v = D3fFeat<FT>::ld4(&f[(size_t)row * ldf + c]);
            s += (double)v.x; s += (double)v.y; s += (double)v.z; s += (double)v.w;
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    if (in && l == 0) pos[row] = s > 0.0 ? 1 : 0;
}

template <int LQ, bool FAST, class FT = float>  // lanes per query = Cin / 4; FAST: linear / sum / 15 kernel points (kp_influences_t)
__global__ void __launch_bounds__(256)
kpconv_agg_vec4(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                int ld_idx, int K, const FT* __restrict__ f, int ldf, const unsigned char* __restrict__ rowpos,
                KpParams P, float* __restrict__ wf, float* __restrict__ inv_cnt, const int* __restrict__ Nq_dev,
                const int* __restrict__ Ns_dev, const int* __restrict__ q_order) {
    constexpr int TQ = 256 / LQ;  // queries per workgroup
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const KpFeatBuf<FT> fbuf(f, Ns, ldf);          // feature rows as a buffer resource (kp_gather4)
    if ((int)(blockIdx.x * TQ) >= Nq) return;   // capacity-sized grid: whole block beyond the real query count
    const int tile = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((Nq + TQ - 1) / TQ));   // one contiguous run of tiles per XCD
    constexpr int KC = LQ;        // neighbours per chunk (TQ*KC = 256 pairs = one per thread)
    constexpr int WS = KC * 16 + 4;  // per-query stride in floats (+4: de-phase the b128 broadcasts of adjacent queries)
    __shared__ __attribute__((aligned(16))) float lw[TQ * WS];
    __shared__ int lidx[TQ * KC];
    __shared__ int lcnt[TQ];
    const int tid = threadIdx.x;
    const int ql = tid / LQ, cl = tid % LQ;  // query-in-block, channel group
    // q_order: a spatially coherent visiting order (cell-sorted): the TQ queries of a workgroup then share most of their
    // neighbours, whose feature rows are fetched once into L1 / L2 instead of TQ times from all over the cloud
    const int qslot = tile * TQ + ql;
    const int qg = (q_order && qslot < Nq) ? q_order[qslot] : qslot;
    const int Cin = LQ * 4;
    if (tid < TQ) lcnt[tid] = 0;
    float acc[KP_MAXP - 1][4];
#pragma unroll
    for (int p = 0; p < KP_MAXP - 1; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.f;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qg < Nq) { qx = q[3 * (size_t)qg]; qy = q[3 * (size_t)qg + 1]; qz = q[3 * (size_t)qg + 2]; }
    const int* idrow = idx + (qg < Nq ? __umul24((unsigned)qg, (unsigned)ld_idx) : 0u);   // (rows, leading dimensions < 2^24: kp_fits_u24)
    KpPair pr = kp_pair_fetch(kp_pair_index(idrow, qg < Nq, cl, K, Ns), Ns, s, rowpos);     // chunk 0's pair
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int id_next = kp_pair_index(idrow, qg < Nq, k0 + KC + cl, K, Ns);              // in flight during phase A
        // ---- phase A: thread = (query ql, neighbour k0 + cl) ----
        {
            float w[KP_MAXP];
            const bool positive = kp_pair_influences<FAST>(P, pr, qx, qy, qz, w);
            kp_count_positive<LQ>(positive, ql, cl, lcnt);
            lidx[ql * KC + cl] = pr.id;
            kp_store_w(&lw[ql * WS + cl * 16], cl, w);
        }
        __syncthreads();
        pr = kp_pair_fetch(id_next, Ns, s, rowpos);                                          // in flight during phase B
        // ---- phase B: thread = (query ql, channels 4*cl .. 4*cl+3) ----
        const int kend = min(KC, K - k0);
        // feature rows are requested in groups of up to eight before any is consumed: the gathers are independent, so their
        // latencies overlap instead of adding up (the kernel is bound by these round trips, not by the FMAs)
        constexpr int PF = KC < 8 ? KC : 8;
        for (int kg = 0; kg < kend; kg += PF) {
            float4 fv[PF];
            int ids[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ids[u] = (kg + u < kend) ? lidx[ql * KC + kg + u] : -1;
                fv[u] = kp_gather4(fbuf, ids[u], ldf, 4 * cl);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (!__any(ids[u] >= 0)) continue;   // (wavefront-uniform) shadow slot for every query of the wavefront: nothing to add
                float w[16];
                kp_load_w(&lw[ql * WS + (kg + u) * 16], kg + u, w);
#pragma unroll
                for (int p = 0; p < KP_MAXP - 1; ++p) {
                    acc[p][0] = fmaf(w[p], fv[u].x, acc[p][0]);
                    acc[p][1] = fmaf(w[p], fv[u].y, acc[p][1]);
                    acc[p][2] = fmaf(w[p], fv[u].z, acc[p][2]);
                    acc[p][3] = fmaf(w[p], fv[u].w, acc[p][3]);
                }
            }
        }
        __syncthreads();
    }
    if (qg < Nq) {
        float* o = wf + (size_t)qg * P.num_kp * Cin + 4 * cl;
#pragma unroll
        for (int p = 0; p < KP_MAXP - 1; ++p)
            if (p < P.num_kp) *(float4*)&o[(size_t)p * Cin] = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        if (cl == 0) inv_cnt[qg] = 1.0f / fmaxf((float)lcnt[ql], 1.0f);
    }
}

// generic path: one thread per (query, channel); every thread recomputes the influences of its query's
// neighbours (free for Cin = 1, the only shipped use: the all-ones input features of layer 0).
__global__ void __launch_bounds__(256)
kpconv_agg_scalar(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                  int ld_idx, int K, const float* __restrict__ f, int ldf, int Cin,
                  const unsigned char* __restrict__ rowpos, KpParams P, float* __restrict__ wf,
                  float* __restrict__ inv_cnt, const int* __restrict__ Nq_dev, const int* __restrict__ Ns_dev) {
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)Nq * Cin) return;
    const int qg = (int)(t / Cin), c = (int)(t % Cin);
    const float qx = q[3 * (size_t)qg], qy = q[3 * (size_t)qg + 1], qz = q[3 * (size_t)qg + 2];
    float acc[KP_MAXP];
#pragma unroll
    for (int p = 0; p < KP_MAXP; ++p) acc[p] = 0.f;
    int cnt = 0;
    for (int k = 0; k < K; ++k) {
        const int id = idx[(size_t)qg * ld_idx + k];
        if (id < 0 || id >= Ns) continue;
        float w[KP_MAXP];
        kp_influences(P, s[3 * (size_t)id] - qx, s[3 * (size_t)id + 1] - qy, s[3 * (size_t)id + 2] - qz, w);
        const float fv = f[(size_t)id * ldf + c];
        cnt += rowpos[id] ? 1 : 0;
#pragma unroll
        for (int p = 0; p < KP_MAXP; ++p) acc[p] = fmaf(w[p], fv, acc[p]);
    }
    for (int p = 0; p < P.num_kp; ++p) wf[((size_t)qg * P.num_kp + p) * Cin + c] = acc[p];
    if (c == 0) inv_cnt[qg] = 1.0f / fmaxf((float)cnt, 1.0f);
}

// ------------------------------------------------------------------------------------------------
// Input layer (Cin = 1, models/network_blocks.py:222-244 `simple_block` on the all-ones features): whole KPConv_ops
// in ONE kernel.  One wavefront per query, lanes = neighbour slots: every lane computes the 15 influences of its
// neighbour, the wave all-reduces the 15 weighted sums (xor butterflies), then lanes switch roles to output channels:
// lane o does the 15-term contraction with K_values[:,0,o] (held in registers across the wave's queries), the
// neighbour-count division and the fused batch-norm / LeakyReLU epilogue, and the 64 lanes store one coalesced row.
// Nothing but the [Nq, Cout] result touches HBM.
// ------------------------------------------------------------------------------------------------
struct KpEpi {
    const float* col_scale;
    const float* col_shift;
    const float* residual;
    int ldr;
    int leaky;
    float alpha;
};

#define C1_QPW 8  // queries per wavefront (amortises the K_values registers)

__global__ void __launch_bounds__(256)
kpconv_c1_fused_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                       int ld_idx, int K, const float* __restrict__ f, int ldf, KpParams P, const float* __restrict__ W,
                       int Cout, KpEpi E, float* __restrict__ out, int ldo, const int* __restrict__ Nq_dev,
                       const int* __restrict__ Ns_dev) {
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int q0 = wave * C1_QPW;
    if (q0 >= Nq) return;
    for (int o0 = 0; o0 < Cout; o0 += 64) {
        const int o = o0 + lane;
        float wreg[KP_MAXP - 1];
#pragma unroll
        for (int p = 0; p < KP_MAXP - 1; ++p) wreg[p] = (p < P.num_kp && o < Cout) ? W[(size_t)p * Cout + o] : 0.f;
        const float cs = (E.col_scale && o < Cout) ? E.col_scale[o] : 1.f;
        const float ch = (E.col_shift && o < Cout) ? E.col_shift[o] : 0.f;
        for (int qi = q0; qi < min(q0 + C1_QPW, Nq); ++qi) {
            const float qx = q[3 * (size_t)qi], qy = q[3 * (size_t)qi + 1], qz = q[3 * (size_t)qi + 2];
            float acc[KP_MAXP - 1];
#pragma unroll
            for (int p = 0; p < KP_MAXP - 1; ++p) acc[p] = 0.f;
            float cnt = 0.f;
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                const int id = (k < K) ? idx[(size_t)qi * ld_idx + k] : -1;
                if (id >= 0 && id < Ns) {
                    float w[KP_MAXP];
                    kp_influences(P, s[3 * (size_t)id] - qx, s[3 * (size_t)id + 1] - qy, s[3 * (size_t)id + 2] - qz, w);
                    const float fv = f[(size_t)id * ldf];
                    cnt += (fv > 0.f) ? 1.f : 0.f;
#pragma unroll
                    for (int p = 0; p < KP_MAXP - 1; ++p) acc[p] = fmaf(w[p], fv, acc[p]);
                }
            }
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) {
                cnt += __shfl_xor(cnt, sh, 64);
#pragma unroll
                for (int p = 0; p < KP_MAXP - 1; ++p) acc[p] += __shfl_xor(acc[p], sh, 64);
            }
            float v = 0.f;
#pragma unroll
            for (int p = 0; p < KP_MAXP - 1; ++p) v = fmaf(acc[p], wreg[p], v);
            v = v * (1.0f / fmaxf(cnt, 1.0f)) * cs + ch;
            if (o < Cout) {
                if (E.residual) v += E.residual[(size_t)qi * E.ldr + o];
                if (E.leaky) v = v > 0.f ? v : v * E.alpha;
                out[(size_t)qi * ldo + o] = v;
            }
        }
    }
}

// Second form of the Cin = 1 kernel, used for aggregation 'sum': lanes = (query, kernel point) instead of (neighbour).
// A wavefront owns 4 queries x 16 lanes; lane p of a query walks that query's neighbours and accumulates the influence of ITS
// kernel point only (lane 15 counts the positive neighbours), so no cross-lane reduction of 15 sums per query is needed
// (the neighbour-lane form spends 90 shuffles per query on it).  The 15-term contraction with K_values[:,0,:] then runs over
// an LDS copy of the 16 x 16 sums with lanes = output channels.
#define C1_SC 64                    // neighbours staged per pass (4 per loader lane)
#define C1_QS (C1_SC * 4 + 4)      // floats per query in LDS; +4 de-phases the four queries of a wavefront (b128 broadcasts)
template <class OT = float>   // OT: output feature storage type (the input feature is the constant-1 column, fp32)
__global__ void __launch_bounds__(256)
kpconv_c1_kp_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                    int ld_idx, int K, const float* __restrict__ f, int ldf, KpParams P, const float* __restrict__ W,
                    int Cout, KpEpi E, OT* __restrict__ out, int ldo, const int* __restrict__ Nq_dev,
                    const int* __restrict__ Ns_dev, const int* __restrict__ q_order) {
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    if ((int)(blockIdx.x * 16) >= Nq) return;
    const int tile = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((Nq + 15) / 16));   // one contiguous run of tiles per XCD
    __shared__ float skp[KP_MAXP * 3];
    __shared__ float sacc[16][17];
    __shared__ int sq[16];
    __shared__ __attribute__((aligned(16))) float snb[16 * C1_QS];     // per query: C1_SC records {s - q, f}
    const int tid = threadIdx.x, p = tid & 15, ql = tid >> 4;
    if (tid < KP_MAXP * 3) skp[tid] = P.kp[tid];
    const int qslot = tile * 16 + ql;
    const bool live = qslot < Nq;
    const int qi = live ? (q_order ? q_order[qslot] : qslot) : 0;
    if (p == 0) sq[ql] = live ? qi : -1;
    __syncthreads();
    const float kx = skp[3 * p], ky = skp[3 * p + 1], kz = skp[3 * p + 2];
    const bool kp_lane = p < P.num_kp;
    const float qx = q[3 * (size_t)qi], qy = q[3 * (size_t)qi + 1], qz = q[3 * (size_t)qi + 2];
    const int* row = idx + (size_t)qi * ld_idx;
    float acc = 0.f;       // lanes 0..14: sum_k h_p * f  (lane 15 walks along with a kernel point of zeros; its sum is dropped)
    float cnt = 0.f;       // number of neighbours with f > 0 (the same number in the 16 lanes of a query)
    const float sig = P.extent * 0.3f, gden = 2.0f * sig * sig + 1e-9f;
    // The 16 lanes of a query first act as LOADERS: lane p fetches neighbours p, p + 16, p + 32, p + 48 of the pass (index,
    // then position and feature, all four in flight) and parks (s - q, f) as one 16-byte record in LDS; then every lane walks
    // the records (one broadcast ds_read_b128 per neighbour).  Before, every lane issued its own four dword loads per
    // neighbour -- 16 identical addresses per instruction -- and the kernel was bound by the CU's address unit, not by bytes.
    float4* mynb = (float4*)(snb + ql * C1_QS);
    const int gshift = 16 * (ql & 3);                                  // this query's lanes inside the wavefront ballot
    for (int k0 = 0; k0 < K; k0 += C1_SC) {
        if (k0) __syncthreads();                                       // the previous pass is consumed by every lane
        int id[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + p + 16 * j;
            id[j] = (live && k < K) ? row[k] : -1;
        }
        float px[4], py[4], pz[4], fv[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ok[j] = id[j] >= 0 && id[j] < Ns;
            const size_t o3 = 3 * (size_t)(ok[j] ? id[j] : 0);
            px[j] = s[o3]; py[j] = s[o3 + 1]; pz[j] = s[o3 + 2];
            fv[j] = ok[j] ? f[(size_t)id[j] * ldf] : 0.f;
        }
        // (round 5) the walk below has no branches: a slot that holds no neighbour carries f = 0 (its influence, computed from
        // support 0's position, is finite in every mode and multiplies nothing), and the positive neighbours are counted HERE, by
        // one ballot per loaded group, instead of by a sixteenth lane that took its own path through every iteration
        // records of TWO neighbours, interleaved {xa xb | ya yb | za zb | fa fb} (round 6): the walk evaluates a pair per step with
        // packed fp32 instructions (v_pk_add / v_pk_mul: the same IEEE result per element, half the issue slots -- the kernel is
        // vector-issue bound, tools/pmc_issue.sh: 0.87); the accumulation stays one fma per neighbour in neighbour order
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* d = (float*)mynb + ((p + 16 * j) >> 1) * 8 + (p & 1);
            d[0] = px[j] - qx; d[2] = py[j] - qy; d[4] = pz[j] - qz; d[6] = fv[j];
            cnt += (float)__popcll((__ballot(ok[j] && fv[j] > 0.f) >> gshift) & 0xFFFFull);
        }
        __syncthreads();
        const int kn = min(C1_SC, K - k0);
        const int np = (kn + 1) >> 1;          // (a slot beyond kn holds a finite position and f = 0: it adds +0)
        typedef float c1_f2 __attribute__((ext_vector_type(2)));
        if (P.influence == 1) {
            const c1_f2 k2x = {kx, kx}, k2y = {ky, ky}, k2z = {kz, kz}, eps = {1e-10f, 1e-10f}, one = {1.0f, 1.0f};
            const c1_f2 nie = {-P.inv_2extent, -P.inv_2extent};
#pragma unroll 4
            for (int m = 0; m < np; ++m) {
                const float4 r0 = mynb[2 * m], r1 = mynb[2 * m + 1];
                const c1_f2 X = {r0.x, r0.y}, Y = {r0.z, r0.w}, Z = {r1.x, r1.y};
                const c1_f2 dx = X - k2x, dy = Y - k2y, dz = Z - k2z;
                // (fma forms as kp_influences_t of the fused kernels: one rounding per step instead of two)
                const c1_f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)) + eps;
                const c1_f2 sq = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                const c1_f2 t = __builtin_elementwise_fma(sq, nie, one);
                acc = fmaf(fmaxf(t.x, 0.0f), r1.z, acc);
                acc = fmaf(fmaxf(t.y, 0.0f), r1.w, acc);
            }
        } else {
#pragma unroll 2
            for (int u = 0; u < kn; ++u) {
                const float* v = (const float*)mynb + (u >> 1) * 8 + (u & 1);
                const float dx = v[0] - kx, dy = v[2] - ky, dz = v[4] - kz;
                const float d2 = dx * dx + dy * dy + dz * dz;
                const float h = P.influence == 0 ? 1.0f : expf(-d2 / gden);
                acc = fmaf(h, v[6], acc);
            }
        }
    }
    sacc[ql][p] = p == 15 ? cnt : (kp_lane ? acc : 0.f);
    __syncthreads();
    // contraction + epilogue: thread -> output channel o = tid % 64 (+64 per pass), queries tid/64 + 4*i
    for (int o0 = 0; o0 < Cout; o0 += 64) {
        const int o = o0 + (tid & 63);
        float wreg[KP_MAXP - 1];
#pragma unroll
        for (int pp = 0; pp < KP_MAXP - 1; ++pp) wreg[pp] = (pp < P.num_kp && o < Cout) ? W[(size_t)pp * Cout + o] : 0.f;
        const float cs = (E.col_scale && o < Cout) ? E.col_scale[o] : 1.f;
        const float ch = (E.col_shift && o < Cout) ? E.col_shift[o] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qq = (tid >> 6) + 4 * i;
            const int gq = sq[qq];
            if (gq < 0 || o >= Cout) continue;
            float v = 0.f;
#pragma unroll
            for (int pp = 0; pp < KP_MAXP - 1; ++pp) v = fmaf(sacc[qq][pp], wreg[pp], v);
            v = v * (1.0f / fmaxf(sacc[qq][15], 1.0f)) * cs + ch;
            if (E.residual) v += E.residual[(size_t)gq * E.ldr + o];
            if (E.leaky) v = v > 0.f ? v : v * E.alpha;
            D3fFeat<OT>::st1(&out[(size_t)gq * ldo + o], v);
        }
    }
}

extern "C" int d3f_kpconv_fused_c1(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                                   const float* f, int ldf, const float* kp_host, int num_kp, float KP_extent, int influence,
                                   int aggregation, const float* W, int Cout, const float* col_scale,
                                   const float* col_shift, const float* residual, int ldr, int leaky, float alpha,
                                   void* out_, int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order,
                                   int out_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    float* out = (float*)out_;
    if (Nq < 0 || Ns < 0 || K < 0 || ld_idx < K || ldf < 1 || num_kp < 1 || num_kp > KP_MAXP - 1 || influence < 0 ||
        influence > 2 || aggregation < 0 || aggregation > 1 || !(KP_extent > 0.f) || Cout < 1 || ldo < Cout ||
        (residual && ldr < Cout))
        return D3F_ERR_ARG;
    if (Nq == 0) return D3F_OK;
    if (!q || !s || !idx || !f || !kp_host || !W || !out) return D3F_ERR_ARG;
    const KpParams P = kp_make_params(kp_host, num_kp, KP_extent, influence, aggregation);
    KpEpi E{col_scale, col_shift, residual, ldr, leaky, alpha};
    if (out_bf16 && !(aggregation == 0 && num_kp <= 15)) return D3F_ERR_ARG;    // bf16 feature storage: the shipped configuration
    if (aggregation == 0 && num_kp <= 15) {
        if (out_bf16)
            kpconv_c1_kp_kernel<unsigned short><<<d3f_cdiv(Nq, 16), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, P, W, Cout, E,
                                                                                      (unsigned short*)out_, ldo, Nq_dev, Ns_dev, q_order);
        else
        kpconv_c1_kp_kernel<float><<<d3f_cdiv(Nq, 16), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, P, W, Cout, E, out, ldo,
                                                                  Nq_dev, Ns_dev, q_order);
    } else {   // 'closest' needs the arg-min over the kernel points of every neighbour: lanes = neighbours
        const long long waves = d3f_cdiv(Nq, C1_QPW);
        kpconv_c1_fused_kernel<<<d3f_cdiv(waves * 64, 256), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, P, W, Cout,
                                                                             E, out, ldo, Nq_dev, Ns_dev);
    }
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// Whole KPConv_ops (kernels/convolution_ops.py:161-255) + inference epilogue for Cin = Cout = 32 -- the two level-0
// convolutions, which own the largest aggregation tensor of the network (wf = 58 739 x 480 floats = 113 MB written and read
// back).  A workgroup owns 32 queries: phases A / B are kpconv_agg_vec4<8>'s; the 32 x 480 tile of weighted features then
// stays in LDS and is contracted with K_values [480, 32] on the matrix cores (v_mfma_f32_32x32x2_f32, the four wavefronts
// split the 240 k-steps and their partial tiles are summed through LDS), followed by the neighbour-count division,
// batch norm and LeakyReLU.  Only out [Nq, 32] reaches HBM.
// ------------------------------------------------------------------------------------------------
// ---- the contraction of the fused kernels by EXACT operand splitting (round 5; the scheme of gemm_x3.h) ------------------------
// Levels 1 and 2 spend more matrix-pipe time in the 15*Cin-deep contraction than vector time in the aggregation (Cin = 64:
// 240 v_mfma_f32_16x16x4_f32 of 32 cycles per wave and tile against ~5000 cycles of FMAs; Cin = 128: 480).  An fp32 value is three
// bfloat16 planes exactly (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)), a product of two planes is exact in fp32, and
// six of the nine plane products carry everything above 2^-23 of the product: 6 v_mfma_f32_16x16x32_bf16 (32 k, ~17 cycles each)
// do the work of 8 v_mfma_f32_16x16x4_f32 (4 k, 32 cycles each): 2.5x less matrix-pipe time, fp32 in / fp32 out, error of the
// order of the fp32 kernel's (tests/test_gpu_kpconv_x3.py).  The weighted features are split ONCE, when the accumulators are
// written to the LDS tile (three planes [16][256 + 8] bf16, 528-byte rows: the 16 lanes of a fragment read hit 16 distinct 4-bank
// groups); K_values is pre-split once per tensor in the B-fragment order (d3f_kpconv_pack_weights_x3): one coalesced 1 KB
// load per wave, plane and 32-deep k-step.  Non-finite weighted features: Inf - Inf = NaN in the second plane -- non-finite in,
// non-finite out, like the fp32 form (which yields +-Inf where this one yields NaN).
typedef __bf16 kp_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned kx_cvt_pk(float lo, float hi) {     // two fp32 -> two bf16 (RNE), lo in bits 0..15
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// four consecutive-k floats -> the three operand planes (four bf16 = one uint2 each)
__device__ __forceinline__ void kx_split4(float x0, float x1, float x2, float x3, uint2 (&pl)[3]) {
    float v[4] = {x0, x1, x2, x3};
    unsigned p[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        p[0][i] = kx_cvt_pk(a, b);
        a -= __uint_as_float(p[0][i] << 16);
        b -= __uint_as_float(p[0][i] & 0xffff0000u);
        p[1][i] = kx_cvt_pk(a, b);
        a -= __uint_as_float(p[1][i] << 16);
        b -= __uint_as_float(p[1][i] & 0xffff0000u);
        p[2][i] = kx_cvt_pk(a, b);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) pl[s] = make_uint2(p[s][0], p[s][1]);
}
#define KX_KT 256                       // k-values per contraction pass of the split form
#define KX_TS (KX_KT + 8)               // bf16 per tile row (528 bytes)

typedef float kp_f32x16 __attribute__((ext_vector_type(16)));
typedef float kp_f32x4 __attribute__((ext_vector_type(4)));

#define KF_TQ 32
#define KF_LQ 8
#define KF_WS (KF_LQ * 16 + 4)        // phase-A stride per query (floats)
#define KF_HP 8                        // kernel points per contraction pass (the wf tile goes through LDS in two passes)
#define KF_TS (KF_HP * 32 + 1)        // wf tile stride per query: odd, so the MFMA A-fragment column reads hit 32 banks

// contraction + epilogue of the Cin = Cout = 32 kernels: the 60 register accumulators of every (query, channel group) thread
// go through the LDS tile in two passes and are contracted with K_values on the matrix cores (see the header comment)
template <class TileWriter, class FT>   // write_tile(wft, p0, np): this thread's weighted features of kernel points p0 .. p0+np-1 -> LDS tile
__device__ __forceinline__ void kf32_contract_epilogue(TileWriter write_tile, int tid, float* kf_smem, const int* lcnt,
                                                       const int* lq, const KpParams& P, const float* __restrict__ W,
                                                       const KpEpi& E, FT* __restrict__ out, int ldo) {
    float* wft = kf_smem;
    // ---- contraction on the matrix cores: out[32 x 32] = wf[32 x 480] @ W[480 x 32], in two passes of 8 / 7 kernel points:
    //      this thread's weighted features -> LDS tile wft[ql][(p - p0)*32 + 4*cl + j] (k index = p*Cin + c, as K_values),
    //      then every wavefront multiplies its quarter of the pass's k-steps ----
    const int lane = tid & 63, wave = tid >> 6;
    kp_f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p0 = pass * KF_HP;
        const int np = min(P.num_kp - p0, KF_HP);
        if (pass) __syncthreads();                      // the previous pass's tile has been consumed
        write_tile(wft, p0, np);
        __syncthreads();
        if (np <= 0) continue;
        const int ksteps = np * 16;                     // k-steps of 2
        const int per = (ksteps + 3) / 4;
        const int kb = wave * per, ke = min(ksteps, kb + per);
        const float* ap = &wft[(lane & 31) * KF_TS + (lane >> 5)];
        const float* bp = W + ((size_t)p0 * 32 + (lane >> 5)) * 32 + (lane & 31);
        for (int kk = kb; kk < ke; kk += 4) {
            float a4[4], b4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k2 = min(kk + u, ke - 1) * 2;
                a4[u] = ap[k2];
                b4[u] = bp[(size_t)k2 * 32];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (kk + u < ke) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u], c, 0, 0, 0);
        }
    }
    __syncthreads();
    // partial tiles -> LDS (the tile region is free now), sum of the four in wave order, epilogue
    float* red = kf_smem;                                   // [4][32*32]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[wave * 1024 + row * 32 + (lane & 31)] = c[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256, row = e >> 5, o = e & 31;
        const int gq = lq[row];
        if (gq < 0) continue;
        float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        v *= 1.0f / fmaxf((float)lcnt[row], 1.0f);
        if (E.col_scale) v *= E.col_scale[o];
        if (E.col_shift) v += E.col_shift[o];
        if (E.residual) v += E.residual[(size_t)gq * E.ldr + o];
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
        D3fFeat<FT>::st1(&out[(size_t)gq * ldo + o], v);
    }
}

// The same contraction + epilogue in the operand-split form (see the block comment above KX_KT; round 5): the tile goes through
// LDS as three bf16 planes [32][160 + 8] in THREE passes of five kernel points (32 KB, the region's size: 336-byte rows put the 16
// lanes of a fragment read on 16 distinct 4-bank groups), the four wavefronts split a pass's ten 16-deep k-steps (3 + 3 + 2 + 2,
// rotated from pass to pass: 8 + 8 + 7 + 7 over the tile), v_mfma_f32_32x32x16_bf16, six products per step; W = the pre-split
// planes in fragment order (d3f_kpconv_pack_weights_x3 with N = 32: Wx[(step * 3 + plane) * 64 + lane][8]).  48 MFMAs of 32 cycles
// per wave and tile instead of 60 of 64.
#define KF3_HP 5
#define KF3_TS (KF3_HP * 32 + 8)      // bf16 per plane row
template <class Acc, class FT>
__device__ __forceinline__ void kf32_contract_epilogue_x3(Acc& acc, int tid, int ql, int cl, float* kf_smem, const int* lcnt,
                                                          const int* lq, const KpParams& P, const unsigned short* __restrict__ Wx,
                                                          const KpEpi& E, FT* __restrict__ out, int ldo) {
    constexpr int PL = KF_TQ * KF3_TS;                       // bf16 per plane
    static_assert(3 * PL * 2 <= KF_TQ * KF_TS * 4, "the plane tile must fit the region");
    unsigned short* tile3 = (unsigned short*)kf_smem;
    const int lane = tid & 63, wave = tid >> 6;
    kp_f32x16 c0, c1;                                        // two accumulator chains, alternating (the products of a step are independent)
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
    const uint4* bw = (const uint4*)Wx + lane;               // + (step * 3 + plane) * 64
    const unsigned short* ap = tile3 + (lane & 31) * KF3_TS + 8 * (lane >> 5);
    const int npass = (P.num_kp + KF3_HP - 1) / KF3_HP;
#define KF3_BLOAD(B_, ST_)                                                                                          \
    do {                                                                                                            \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) B_[pl] = bw[(size_t)((ST_) * 3 + pl) * 64];                \
    } while (0)
#define KF3_STEP(B_, LS_)                                                                                                      \
    do {                                                                                                                       \
        uint4 a_[3];                                                                                                           \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) a_[pl] = *(const uint4*)(ap + pl * PL + 16 * (LS_));                  \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[2]), __builtin_bit_cast(kp_bf16x8, B_[0]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[0]), c1, 0, 0, 0); \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[1]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[2]), c1, 0, 0, 0); \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[1]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[0]), c1, 0, 0, 0); \
    } while (0)
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (pass >= npass) break;                             // (workgroup-uniform)
        const int p0 = pass * KF3_HP;
        const int np = min(P.num_kp - p0, KF3_HP);
        const int S = 2 * np;                                 // 16-deep steps of the pass
        const int s0 = (wave + 2 * pass) & 3;                 // this wave's steps: s0, s0 + 4, s0 + 8
        uint4 b0[3], b1[3];
        // the W fragments of the wave's first two steps are requested before the tile is written (clamped addresses: a wave with
        // fewer steps multiplies nothing with them), the third while the second is multiplied
        KF3_BLOAD(b0, p0 * 2 + min(s0, S - 1));
        KF3_BLOAD(b1, p0 * 2 + min(s0 + 4, S - 1));
        if (pass) __syncthreads();                            // the previous pass's planes have been consumed
#pragma unroll
        for (int p = 0; p < KP_MAXP - 1; ++p) {
            const int pp = p - p0;
            if (pp >= 0 && pp < KF3_HP && p < P.num_kp) {
                uint2 pl[3];
                kx_split4(acc[p][0], acc[p][1], acc[p][2], acc[p][3], pl);
                unsigned short* d = tile3 + ql * KF3_TS + pp * 32 + 4 * cl;
                *(uint2*)d = pl[0];
                *(uint2*)(d + PL) = pl[1];
                *(uint2*)(d + 2 * PL) = pl[2];
            }
        }
        __syncthreads();
        if (s0 < S) KF3_STEP(b0, s0);
        if (s0 + 8 < S) KF3_BLOAD(b0, p0 * 2 + s0 + 8);
        if (s0 + 4 < S) KF3_STEP(b1, s0 + 4);
        if (s0 + 8 < S) KF3_STEP(b0, s0 + 8);
    }
#undef KF3_STEP
#undef KF3_BLOAD
    __syncthreads();
    // partial tiles -> LDS (the tile region is free now), sum of the four in wave order, epilogue (as the fp32 form)
    float* red = kf_smem;                                   // [4][32*32]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[wave * 1024 + row * 32 + (lane & 31)] = c0[r] + c1[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256, row = e >> 5, o = e & 31;
        const int gq = lq[row];
        if (gq < 0) continue;
        float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        v *= 1.0f / fmaxf((float)lcnt[row], 1.0f);
        if (E.col_scale) v *= E.col_scale[o];
        if (E.col_shift) v += E.col_shift[o];
        if (E.residual) v += E.residual[(size_t)gq * E.ldr + o];
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
        D3fFeat<FT>::st1(&out[(size_t)gq * ldo + o], v);
    }
}

template <bool FAST, int PF = 8, class FT = float, bool X3 = false>   // PF: feature rows requested before any is consumed (4: 128 registers, four workgroups per CU)
__global__ void __launch_bounds__(256, PF == 4 ? 4 : 3)
kpconv_fused32_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                      int ld_idx, int K, const FT* __restrict__ f, int ldf, const unsigned char* __restrict__ rowpos,
                      KpParams P, const float* __restrict__ W, KpEpi E, FT* __restrict__ out, int ldo,
                      const int* __restrict__ Nq_dev, const int* __restrict__ Ns_dev, const int* __restrict__ q_order) {
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const KpFeatBuf<FT> fbuf(f, Ns, ldf);          // feature rows as a buffer resource (kp_gather4)
    if ((int)(blockIdx.x * KF_TQ) >= Nq) return;
    const int tile = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((Nq + KF_TQ - 1) / KF_TQ));   // one contiguous run of tiles per XCD
    // One LDS region serves three lives: phase-A influences [32][132] during the neighbour loop, then the weighted-feature
    // tile [32][257] (8 kernel points at a time: MFMA A operand), then the four partial out tiles.  34 KB per workgroup
    // instead of 80 KB: the kernel is bound by dependent gathers, so resident workgroups per CU are what it needs.
    extern __shared__ __attribute__((aligned(16))) float kf_smem[];
    float* wft = kf_smem;
    float* lw = kf_smem;
    int* lidx = (int*)(kf_smem + KF_TQ * KF_TS);            // [32][8]
    int* lcnt = lidx + KF_TQ * KF_LQ;                       // [32]
    int* lq = lcnt + KF_TQ;                                 // [32] global query index of each tile row
    const int tid = threadIdx.x;
    const int ql = tid / KF_LQ, cl = tid % KF_LQ;
    const int qslot = tile * KF_TQ + ql;
    const int qg = (q_order && qslot < Nq) ? q_order[qslot] : qslot;
    if (tid < KF_TQ) lcnt[tid] = 0;
    if (cl == 0) lq[ql] = qslot < Nq ? qg : -1;
    float acc[KP_MAXP - 1][4];
#pragma unroll
    for (int p = 0; p < KP_MAXP - 1; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.f;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qslot < Nq) { qx = q[3 * (size_t)qg]; qy = q[3 * (size_t)qg + 1]; qz = q[3 * (size_t)qg + 2]; }
    const int* idrow = idx + (qslot < Nq ? __umul24((unsigned)qg, (unsigned)ld_idx) : 0u);   // (rows, leading dimensions < 2^24)
    KpPair pr = kp_pair_fetch(kp_pair_index(idrow, qslot < Nq, cl, K, Ns), Ns, s, rowpos);  // chunk 0's pair
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += KF_LQ) {
        const int id_next = kp_pair_index(idrow, qslot < Nq, k0 + KF_LQ + cl, K, Ns);       // in flight during phase A
        {   // ---- phase A: thread = (query ql, neighbour k0 + cl) ----
            float w[KP_MAXP];
            const bool positive = kp_pair_influences<FAST>(P, pr, qx, qy, qz, w);
            kp_count_positive<KF_LQ>(positive, ql, cl, lcnt);
            lidx[ql * KF_LQ + cl] = pr.id;
            kp_store_w(&lw[ql * KF_WS + cl * 16], cl, w);
        }
        __syncthreads();
        pr = kp_pair_fetch(id_next, Ns, s, rowpos);                                         // in flight during phase B
        // ---- phase B: thread = (query ql, channels 4*cl .. 4*cl+3) ----
        // all eight feature rows of the chunk are requested before any is consumed: the gathers are independent, so their
        // latencies overlap instead of adding up (the kernel is bound by these round trips, not by the FMAs)
#pragma unroll
        for (int k1 = 0; k1 < KF_LQ; k1 += PF) {
            float4 fv[PF];
            int ids[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ids[u] = lidx[ql * KF_LQ + k1 + u];
                fv[u] = kp_gather4(fbuf, ids[u], ldf, 4 * cl);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (!__any(ids[u] >= 0)) continue;   // (wavefront-uniform) shadow slot for every query of the wavefront: nothing to add
                float w[16];
                kp_load_w(&lw[ql * KF_WS + (k1 + u) * 16], k1 + u, w);
#pragma unroll
                for (int p = 0; p < KP_MAXP - 1; ++p) {
                    acc[p][0] = fmaf(w[p], fv[u].x, acc[p][0]);
                    acc[p][1] = fmaf(w[p], fv[u].y, acc[p][1]);
                    acc[p][2] = fmaf(w[p], fv[u].z, acc[p][2]);
                    acc[p][3] = fmaf(w[p], fv[u].w, acc[p][3]);
                }
            }
        }
        __syncthreads();
    }
    auto write_tile = [&](float* wft, int p0, int np) {
#pragma unroll
        for (int pp = 0; pp < KF_HP; ++pp) {
            const int p = p0 + pp;
            if (p < KP_MAXP - 1 && pp < np) {
                float* d = &wft[ql * KF_TS + pp * 32 + 4 * cl];
                d[0] = acc[p][0]; d[1] = acc[p][1]; d[2] = acc[p][2]; d[3] = acc[p][3];
            }
        }
    };
    if constexpr (X3) kf32_contract_epilogue_x3(acc, tid, ql, cl, kf_smem, lcnt, lq, P, (const unsigned short*)W, E, out, ldo);
    else kf32_contract_epilogue(write_tile, tid, kf_smem, lcnt, lq, P, W, E, out, ldo);
}

template <bool X3>
static int kp_fused32_launch(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                             const void* f_, int ldf, const unsigned char* rowpos, const float* kp_host, int num_kp,
                             float KP_extent, int influence, int aggregation, const float* W, const float* col_scale,
                             const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* out_,
                             int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order, int feat_bf16,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* f = (const float*)f_;
    float* out = (float*)out_;
    if (Nq < 0 || Ns < 0 || K < 0 || ld_idx < K || ldf < 32 || (ldf % 4) || num_kp < 1 || num_kp > KP_MAXP - 1 ||
        influence < 0 || influence > 2 || aggregation < 0 || aggregation > 1 || !(KP_extent > 0.f) || ldo < 32 ||
        (residual && ldr < 32))
        return D3F_ERR_ARG;
    if (Nq == 0) return D3F_OK;
    if (!q || !s || !idx || !f || !rowpos || !kp_host || !W || !out || (((uintptr_t)f) & 15)) return D3F_ERR_ARG;
    if (!kp_fits_u24(Nq, Ns, ld_idx, ldf)) return D3F_ERR_ARG;     // (16.7 M rows per call: use d3f_kpconv_aggregate + d3f_gemm_f32)
    const KpParams P = kp_make_params(kp_host, num_kp, KP_extent, influence, aggregation);
    KpEpi E{col_scale, col_shift, residual, ldr, leaky, alpha};
    const size_t lds = (size_t)(KF_TQ * KF_TS) * sizeof(float) + (size_t)(KF_TQ * KF_LQ + 2 * KF_TQ) * sizeof(int);
    static_assert(KF_TQ * KF_TS >= KF_TQ * KF_WS && KF_TQ * KF_TS >= 4096, "LDS region must hold every life");
    static std::atomic<unsigned long long> lds_done{0};
    // feature rows requested before any is consumed: 4 -> 126 / 123 registers = four workgroups per CU.  (Round 1 measured the
    // deeper prefetch faster -- then 4 spilled; since the packed influences and 24-bit addressing of round 3 it fits and wins:
    // 1542 / 1535 against 1521 / 1529 fragments/s, profiles/r03_experiments.txt x16.)
#define D3F_KP_PF 4
#define D3F_KP_PF_H 4          // the same for bf16 feature rows (120 registers): 202 -> 178 us and 112 -> 99 us per launch (x24)
    const void* const fns[2] = {(const void*)kpconv_fused32_kernel<true, D3F_KP_PF, float, X3>,
                                (const void*)kpconv_fused32_kernel<false, 8, float, X3>};
    if (d3f_opt_in_lds(lds_done, fns, (int)lds) != D3F_OK) return D3F_ERR_HIP;
#define D3F_KF(FAST_, PF_)                                                                                                   \
    kpconv_fused32_kernel<FAST_, PF_, float, X3><<<d3f_cdiv(Nq, KF_TQ), 256, lds, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, rowpos, P, \
                                                                                            W, E, out, ldo, Nq_dev, Ns_dev, q_order)
    if (feat_bf16) {      // bf16 feature storage (in and out); the shipped configuration only; no residual operand
        if (!kp_fast_config(num_kp, influence, aggregation) || residual) return D3F_ERR_ARG;
        static std::atomic<unsigned long long> lds_done_h{0};
        const void* const fnh[1] = {(const void*)kpconv_fused32_kernel<true, D3F_KP_PF_H, unsigned short, X3>};
        if (d3f_opt_in_lds(lds_done_h, fnh, (int)lds) != D3F_OK) return D3F_ERR_HIP;
        kpconv_fused32_kernel<true, D3F_KP_PF_H, unsigned short, X3><<<d3f_cdiv(Nq, KF_TQ), 256, lds, stream>>>(
            q, Nq, s, Ns, idx, ld_idx, K, (const unsigned short*)f_, ldf, rowpos, P, W, E, (unsigned short*)out_, ldo, Nq_dev, Ns_dev, q_order);
    } else if (!kp_fast_config(num_kp, influence, aggregation)) D3F_KF(false, 8);
    else D3F_KF(true, D3F_KP_PF);
#undef D3F_KF
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
// ------------------------------------------------------------------------------------------------
// Cin = Cout = 32 with the AGGREGATION on the matrix cores too (round 6).
// kpconv_fused32_kernel is bound by vector issue: per 8-neighbour chunk and wavefront ~95 instructions of influence arithmetic and
// ~380 for the 480 multiply-adds per lane (60 accumulators x 8 neighbours as v_pk_fma, plus the LDS broadcasts of the influences).
// wf[q] = W_q^T F_q is a [15 x K] x [K x 32] product per query -- v_mfma_f32_16x16x1_4b_f32 does FOUR such rank-1 updates per
// instruction (one per 16-lane block), in exact fp32 and in the same neighbour order as the FMA chains (the instruction IS an fmaf
// chain over k): lanes 16 b .. 16 b + 15 of a wavefront belong to query b of a group of four; as the A operand lane i supplies the
// influence of kernel point i on the step's neighbour (one ds_read_b32 from phase A's records), as the B operands channels 2 i and
// 2 i + 1 of the neighbour's feature row (one 8-byte buffer load; shadows read zeros through the range check).  Two accumulator
// chains of 16 registers per group; the matrix pipe does the 17.8 k multiply-adds of a query in ~600 cycles while the vector pipe
// computes the next chunk's influences: the two halves of the old kernel's 1190 cycles per query overlap instead of adding up.
//   * phase A as before: thread (query, neighbour of the chunk) -> 16 influences in LDS.  The eight queries of a wavefront's
//     phase B are exactly the queries its own lanes served in phase A: NO workgroup barrier inside the neighbour loop.
//   * D layout (tools/ubench/mfma_16x16x1_layout.hip, checked on the hardware): register r of lane l holds query r / 4 of the group,
//     kernel point 4 (l / 16) + r % 4, channel 2 (l % 16) + chain.  The weighted-feature tile goes to LDS with k' = 16 c + p
//     (channel-major, 16 kernel-point slots, the 16th with zero weights: K' = 512): a lane's four registers of a query are four
//     consecutive k' -> one 8-byte store per plane.  Four passes of 8 channels (128 k', 26 KB for the three bf16 planes);
//   * the contraction as in kf32_contract_epilogue_x3: v_mfma_f32_32x32x16_bf16 on the three-plane split, each wavefront two of a
//     pass's eight 16-deep steps (48 MFMAs per wavefront and tile), W pre-split in fragment order over k' (the packed copy of
//     K_values permuted to [c][16][n]: d3f_kpconv_pack_weights_x3 of that matrix).
// The shipped configuration only (15 kernel points, linear influence, sum); fp32 features.
// ------------------------------------------------------------------------------------------------
#define KM_PC 16                         // channels per contraction pass: the even ones (chain 0), then the odd ones (chain 1)
#define KM_KT (KM_PC * 16)               // k' values per pass
#define KM_TS (KM_KT + 8)                // bf16 per plane row (528 bytes: 16 lanes of a fragment read on 16 distinct 4-bank groups)
#define KM_REGION (3 * KF_TQ * KM_TS * 2)   // bytes of the three planes of a pass: the largest life of the LDS region
__global__ void __launch_bounds__(256, 3)
kpconv_fused32m_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                       int ld_idx, int K, const float* __restrict__ f, int ldf, const unsigned char* __restrict__ rowpos,
                       KpParams P, const unsigned short* __restrict__ Wx, KpEpi E, float* __restrict__ out, int ldo,
                       const int* __restrict__ Nq_dev, const int* __restrict__ Ns_dev, const int* __restrict__ q_order) {
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const KpFeatBuf<float> fbuf(f, Ns, ldf);
    if ((int)(blockIdx.x * KF_TQ) >= Nq) return;
    const int tile = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((Nq + KF_TQ - 1) / KF_TQ));
    // one LDS region, three lives: the influence records [32][132] floats of the neighbour loop, then the three bf16 planes
    // [32][136] of a contraction pass, then the four partial out tiles
    constexpr int PL = KF_TQ * KM_TS;                        // bf16 per plane
    static_assert(2 * KF_TQ * KF_WS * 4 <= KM_REGION && 4 * 1024 * 4 <= KM_REGION && KM_REGION % 16 == 0, "region");
    extern __shared__ __attribute__((aligned(16))) float kf_smem[];
    float* lw = kf_smem;
    int* lidx = (int*)(kf_smem + KM_REGION / 4);             // [2][32][8]   (behind the region)
    int* lcnt = lidx + 2 * KF_TQ * KF_LQ;                    // [32]
    int* lq = lcnt + KF_TQ;                                  // [32] global query index of each tile row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = tid / KF_LQ, cl = tid % KF_LQ;            // phase A: (query of the tile, neighbour of the chunk)
    const int qslot = tile * KF_TQ + ql;
    const int qg = (q_order && qslot < Nq) ? q_order[qslot] : qslot;
    if (tid < KF_TQ) lcnt[tid] = 0;
    if (cl == 0) lq[ql] = qslot < Nq ? qg : -1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qslot < Nq) { qx = q[3 * (size_t)qg]; qy = q[3 * (size_t)qg + 1]; qz = q[3 * (size_t)qg + 2]; }
    const int* idrow = idx + (qslot < Nq ? __umul24((unsigned)qg, (unsigned)ld_idx) : 0u);
    // every index of the tile up front, through LDS (32 rows x K <= 64): the loop below then depends on no load it has just issued
    // -- one index round trip per tile instead of one per chunk on every wavefront's critical path; the support points of the
    // first two chunks right behind
    int* lall = lq + KF_TQ;                                  // [32][64]
    for (int e = tid; e < KF_TQ * 64; e += 256) {
        const int r = e >> 6, k = e & 63;
        const int qs = tile * KF_TQ + r;
        int v = Ns;
        if (qs < Nq && k < K) {
            const int g = q_order ? q_order[qs] : qs;
            v = idx[__umul24((unsigned)g, (unsigned)ld_idx) + k];
        }
        lall[e] = v;
    }
    __syncthreads();
    KpPair pr = kp_pair_fetch(lall[ql * 64 + cl], Ns, s, rowpos);                        // chunk 0's pair
    KpPair pr2 = kp_pair_fetch(lall[ql * 64 + KF_LQ + cl], Ns, s, rowpos);               // chunk 1's pair
    // phase B: lane = (query b of the group, kernel point / channel pair i)
    const int b = lane >> 4, i = lane & 15;
    int aoff[4];                                             // position of element i inside a record whose quads are rotated by 0..3
#pragma unroll
    for (int r = 0; r < 4; ++r) aoff[r] = ((((i >> 2) + r) & 3) << 2) + (i & 3);
    kp_f32x16 acc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][0][r] = acc[g][1][r] = 0.f;
    __syncthreads();                                         // lcnt / lq
    // The records (influences, row offsets) are double buffered: phase A of chunk c + 1 runs -- vector work -- while the feature rows
    // of chunk c are on their way and before its rank-1 updates occupy the matrix pipe; nothing in the loop waits for a round trip
    // it has just started.
    constexpr int LWB = KF_TQ * KF_WS;                       // floats per record buffer
    auto phase_a = [&](int buf, int k0_next_index) {
        float w[KP_MAXP];
        const bool positive = kp_pair_influences<true>(P, pr, qx, qy, qz, w);
        kp_count_positive<KF_LQ>(positive, ql, cl, lcnt);
        // the byte offset of the neighbour's feature row, once per pair (a shadow: beyond the buffer -> the range check returns
        // zeros), instead of a multiply per lane and step in phase B
        lidx[buf * (KF_TQ * KF_LQ) + ql * KF_LQ + cl] = pr.id >= 0 ? (int)(__umul24((unsigned)pr.id, (unsigned)ldf) * 4u) : (int)0xfffffff0u;
        kp_store_w(&lw[buf * LWB + ql * KF_WS + cl * 16], cl, w);
        (void)k0_next_index;
    };
    phase_a(0, 0);
    pr = pr2;
    for (int c = 0, k0 = 0; k0 < K; ++c, k0 += KF_LQ) {
        const int buf = c & 1;
        // (the records this wavefront reads were written by its own lanes: wavefront-level ordering is enough)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool more = k0 + KF_LQ < K;
        // the pair of chunk c + 2: in flight over this chunk's loads and updates and the next chunk's phase A
        if (k0 + 2 * KF_LQ < K) pr2 = kp_pair_fetch(lall[ql * 64 + k0 + 2 * KF_LQ + cl], Ns, s, rowpos);
        // ---- the chunk's feature rows: all sixteen steps (two groups of four queries x eight neighbours) requested at once ----
        unsigned offs[2][KF_LQ];
        float2 fv[2][KF_LQ];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int row = wave * 8 + g * 4 + b;            // tile row of this lane's query
#pragma unroll
            for (int k1 = 0; k1 < KF_LQ; ++k1) {
                offs[g][k1] = (unsigned)lidx[buf * (KF_TQ * KF_LQ) + row * KF_LQ + k1];
                typedef unsigned kp_u2 __attribute__((ext_vector_type(2)));
                const unsigned off = offs[g][k1] == 0xfffffff0u ? 0xfffffff0u : offs[g][k1] + 8u * (unsigned)i;   // (lane offset on real rows only)
                const kp_u2 v = __builtin_amdgcn_raw_buffer_load_b64(fbuf.r, (int)off, 0, 0);
                fv[g][k1] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
            }
        }
        if (more) {
            phase_a(buf ^ 1, 0);                             // ---- phase A of the NEXT chunk, under the loads ----
            pr = pr2;
        }
        // ---- phase B: one rank-1 update per neighbour and chain, four queries per instruction ----
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int row = wave * 8 + g * 4 + b;
            float av[KF_LQ];
#pragma unroll
            for (int k1 = 0; k1 < KF_LQ; ++k1)               // the record's quads are rotated by slot / 2 (kp_store_w)
                av[k1] = lw[buf * LWB + row * KF_WS + k1 * 16 + aoff[k1 >> 1]];
#pragma unroll
            for (int k1 = 0; k1 < KF_LQ; ++k1) {
                if (!__any(offs[g][k1] != 0xfffffff0u)) continue;   // (wavefront-uniform) a shadow slot for all four queries
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x1f32(av[k1], fv[g][k1].x, acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x1f32(av[k1], fv[g][k1].y, acc[g][1], 0, 0, 0);
            }
        }
    }
    // ---- contraction: two passes of 16 channels -- the even ones (chain 0), then the odd ones (chain 1): EVERY lane writes the
    //      registers of one chain per pass (a pass by lane subsets issued every store four times with a quarter of the lanes) ----
    unsigned short* tile3 = (unsigned short*)kf_smem;
    kp_f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
    const uint4* bw = (const uint4*)Wx + lane;               // + (step * 3 + plane) * 64
    const unsigned short* ap = tile3 + (lane & 31) * KM_TS + 8 * (lane >> 5);
#define KM_BLOAD(B_, ST_)                                                                                          \
    do {                                                                                                           \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) B_[pl] = bw[(size_t)((ST_) * 3 + pl) * 64]; \
    } while (0)
#define KM_STEP(B_, LS_)                                                                                                       \
    do {                                                                                                                       \
        uint4 a_[3];                                                                                                           \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) a_[pl] = *(const uint4*)(ap + pl * PL + 16 * (LS_));                  \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[2]), __builtin_bit_cast(kp_bf16x8, B_[0]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[0]), c1, 0, 0, 0); \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[1]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[2]), c1, 0, 0, 0); \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[1]), c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[0]), c1, 0, 0, 0); \
    } while (0)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // this wavefront's four 16-deep steps of the pass: local steps wave, wave + 4, + 8, + 12 (a step = one channel's 16 slots)
        uint4 b0[3], b1[3];
        KM_BLOAD(b0, pass * 16 + wave);
        KM_BLOAD(b1, pass * 16 + wave + 4);
        __syncthreads();                                     // the region's previous life (records / the last pass's planes) is over
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 pl[3];
                kx_split4(acc[g][pass][4 * t], acc[g][pass][4 * t + 1], acc[g][pass][4 * t + 2], acc[g][pass][4 * t + 3], pl);
                // row = query 8 wave + 4 g + t; column = 16 i (this lane's channel 2 i + pass) + 4 b: kernel points 4 b .. 4 b + 3
                unsigned short* d = tile3 + (wave * 8 + g * 4 + t) * KM_TS + i * 16 + 4 * b;
                *(uint2*)d = pl[0];
                *(uint2*)(d + PL) = pl[1];
                *(uint2*)(d + 2 * PL) = pl[2];
            }
        __syncthreads();
        KM_STEP(b0, wave);
        KM_BLOAD(b0, pass * 16 + wave + 8);
        KM_STEP(b1, wave + 4);
        KM_BLOAD(b1, pass * 16 + wave + 12);
        KM_STEP(b0, wave + 8);
        KM_STEP(b1, wave + 12);
    }
#undef KM_STEP
#undef KM_BLOAD
    __syncthreads();
    // partial tiles -> LDS (the region is free now), sum of the four in wave order, epilogue (as the other forms)
    float* red = kf_smem;                                   // [4][32*32]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[wave * 1024 + row * 32 + (lane & 31)] = c0[r] + c1[r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * 256, row = e >> 5, o = e & 31;
        const int gq = lq[row];
        if (gq < 0) continue;
        float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        v *= 1.0f / fmaxf((float)lcnt[row], 1.0f);
        if (E.col_scale) v *= E.col_scale[o];
        if (E.col_shift) v += E.col_shift[o];
        if (E.residual) v += E.residual[(size_t)gq * E.ldr + o];
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
        out[(size_t)gq * ldo + o] = v;
    }
}

// W = d3f_kpconv_pack_weights_x3 of the [512, 32] matrix W'[16 s + p][n] = K_values[p][c(s)][n] (p < 15), 0 (p = 15), with
// c(s) = 2 s for s < 16 (the even channels first), 2 (s - 16) + 1 after
extern "C" int d3f_kpconv_fused32_mfma(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                                       const float* f, int ldf, const unsigned char* rowpos, const float* kp_host, int num_kp,
                                       float KP_extent, int influence, int aggregation, const void* Wx, const float* col_scale,
                                       const float* col_shift, const float* residual, int ldr, int leaky, float alpha, float* out,
                                       int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Nq < 0 || Ns < 0 || K < 0 || ld_idx < K || ldf < 32 || (ldf % 4) || !(KP_extent > 0.f) || ldo < 32 || (residual && ldr < 32))
        return D3F_ERR_ARG;
    if (!kp_fast_config(num_kp, influence, aggregation) || K > 64) return D3F_ERR_ARG;      // (eight chunks of eight neighbours)
    if (Nq == 0) return D3F_OK;
    if (!q || !s || !idx || !f || !rowpos || !kp_host || !Wx || !out || (((uintptr_t)f | (uintptr_t)Wx) & 15)) return D3F_ERR_ARG;
    if (!kp_fits_u24(Nq, Ns, ld_idx, ldf)) return D3F_ERR_ARG;
    const KpParams P = kp_make_params(kp_host, num_kp, KP_extent, influence, aggregation);
    KpEpi E{col_scale, col_shift, residual, ldr, leaky, alpha};
    const size_t lds = (size_t)KM_REGION + (size_t)(2 * KF_TQ * KF_LQ + 2 * KF_TQ + KF_TQ * 64) * sizeof(int);
    kpconv_fused32m_kernel<<<d3f_cdiv(Nq, KF_TQ), 256, lds, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, rowpos, P,
                                                                      (const unsigned short*)Wx, E, out, ldo, Nq_dev, Ns_dev, q_order);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

#define KP_F32_ARGS                                                                                                               \
    const float *q, int Nq, const float *s, int Ns, const int *idx, int ld_idx, int K, const void *f_, int ldf,                   \
        const unsigned char *rowpos, const float *kp_host, int num_kp, float KP_extent, int influence, int aggregation,           \
        const float *W, const float *col_scale, const float *col_shift, const float *residual, int ldr, int leaky, float alpha,   \
        void *out_, int ldo, const int *Nq_dev, const int *Ns_dev, const int *q_order, int feat_bf16, void *stream_
#define KP_F32_PASS                                                                                                               \
    q, Nq, s, Ns, idx, ld_idx, K, f_, ldf, rowpos, kp_host, num_kp, KP_extent, influence, aggregation, W, col_scale, col_shift,   \
        residual, ldr, leaky, alpha, out_, ldo, Nq_dev, Ns_dev, q_order, feat_bf16, stream_
extern "C" int d3f_kpconv_fused32(KP_F32_ARGS) { return kp_fused32_launch<false>(KP_F32_PASS); }
// the same operator with the contraction in the operand-split form; W = d3f_kpconv_pack_weights_x3(K_values [480, 32])'s planes
extern "C" int d3f_kpconv_fused32_x3(KP_F32_ARGS) { return kp_fused32_launch<true>(KP_F32_PASS); }
#undef KP_F32_ARGS
#undef KP_F32_PASS

// ------------------------------------------------------------------------------------------------
// Whole KPConv_ops (kernels/convolution_ops.py:161-255) + inference epilogue for Cin = 64 / 128 -- levels 1 and 2, whose
// aggregation tensors wf = [Nq, 15*Cin] were written to and read back from memory by the two-kernel form (224 MB each way
// per level-1 layer at four fragments per replay, against 41 MB of algorithmic traffic).  Here wf never leaves the chip:
//   * a workgroup owns 16 queries x LQ = Cin/4 lanes (256 / 512 threads); phases A / B are kpconv_agg_vec4<LQ>'s (influences
//     of one (query, neighbour) pair per thread parked in LDS, then 60 register accumulators per lane over float4 feature
//     gathers, eight in flight);
//   * the 16 x (15*Cin) tile of weighted features then goes through ONE 33 KB LDS region in passes of 512 k-values
//     (8 kernel points at Cin = 64, 4 at Cin = 128) and is contracted with K_values on the matrix cores
//     (v_mfma_f32_16x16x4_f32, exact fp32): wave w owns output columns 16w..16w+15 over the whole k range -- no cross-wave
//     reduction -- with two accumulator chains (the instruction's dependent latency is 40 cycles against a 32-cycle issue);
//   * k may be visited in any order, so lane (row r, group g) owns k = 16*blk + 4*g + {0,1,2,3} of every 16-deep block: its
//     four A operands are ONE ds_read_b128, its four B operands ONE 16-byte load from the k-block-packed copy of K_values
//     (d3f_kpconv_pack_weights: Wp[blk][g][n][j] = W[16*blk + 4*g + j][n]; 256 contiguous bytes per 16-lane group), two
//     groups of four blocks prefetched ahead of the multiplies;
//   * neighbour-count division, batch norm, residual and LeakyReLU in the accumulator registers; only out [Nq, Cout]
//     reaches memory.
// VALU (aggregation) and matrix (contraction) phases of different workgroups on a CU overlap: the two pipes are separate.
// ------------------------------------------------------------------------------------------------
#define KG_TQ 16                        // queries per workgroup = rows of one 16x16x4 tile
#define KG_KT 512                       // k-values per contraction pass
#define KG_TS (KG_KT + 4)               // LDS row stride of the wf tile (floats): 16-byte aligned rows, TS/4 odd

// Registers: 60 accumulators + the gather prefetch decide the occupancy.  512-thread workgroups (Cin = 128) are two waves
// per SIMD each: at more than 128 registers only ONE workgroup fits a CU and its gather and matrix phases cannot overlap with
// anybody's, so that variant prefetches four rows instead of eight and is held to 128 registers (two workgroups per CU).
template <int LQ, int PF = (LQ >= 32 ? 4 : 8), class FT = float, bool X3 = false>   // lanes per query = Cin / 4 (16, 32 or 64); Cout == Cin; waves = LQ / 4 = Cout / 16
__global__ void __launch_bounds__(KG_TQ * LQ, PF == 4 ? 4 : 3)                      // X3: Wp = the pre-split planes (d3f_kpconv_pack_weights_x3)
kpconv_fused_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ s, int Ns, const int* __restrict__ idx,
                    int ld_idx, int K, const FT* __restrict__ f, int ldf, const unsigned char* __restrict__ rowpos,
                    KpParams P, const float* __restrict__ Wp, KpEpi E, FT* __restrict__ out, int ldo,
                    const int* __restrict__ Nq_dev, const int* __restrict__ Ns_dev, const int* __restrict__ q_order) {
    constexpr int CIN = 4 * LQ, COUT = CIN;
    constexpr int KC = LQ < 32 ? LQ : 32;   // neighbours per chunk (one (query, neighbour) pair per thread; Cin = 256: the first
                                            // 32 of a query's 64 lanes -- a wider chunk would not fit the tile region)
    constexpr int WS = KC * 16 + 4;         // phase-A stride per query (floats)
    constexpr int HP = KG_KT / CIN;         // kernel points per contraction pass
    static_assert(KG_TQ * WS <= KG_TQ * KG_TS, "the influence region must fit the tile region");
    Nq = d3f_dyn(Nq, Nq_dev);
    Ns = d3f_dyn(Ns, Ns_dev);
    const KpFeatBuf<FT> fbuf(f, Ns, ldf);          // feature rows as a buffer resource (kp_gather4)
    if ((int)(blockIdx.x * KG_TQ) >= Nq) return;
    const int tile = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((Nq + KG_TQ - 1) / KG_TQ));   // one contiguous run of tiles per XCD
    __shared__ __attribute__((aligned(16))) float region[KG_TQ * KG_TS];   // influences, then the wf tile of each pass
    __shared__ int lidx[KG_TQ * KC];
    __shared__ int lcnt[KG_TQ];
    __shared__ int lq[KG_TQ];
    float* lw = region;
    float* wft = region;
    const int tid = threadIdx.x;
    const int ql = tid / LQ, cl = tid % LQ;
    const int qslot = tile * KG_TQ + ql;
    const int qg = (q_order && qslot < Nq) ? q_order[qslot] : qslot;
    if (tid < KG_TQ) lcnt[tid] = 0;
    if (cl == 0) lq[ql] = qslot < Nq ? qg : -1;
    float acc[KP_MAXP - 1][4];
#pragma unroll
    for (int p = 0; p < KP_MAXP - 1; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.f;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qslot < Nq) { qx = q[3 * (size_t)qg]; qy = q[3 * (size_t)qg + 1]; qz = q[3 * (size_t)qg + 2]; }
    const int* idrow = idx + (qslot < Nq ? __umul24((unsigned)qg, (unsigned)ld_idx) : 0u);   // (rows, leading dimensions < 2^24)
    const bool pair_lane = cl < KC && qslot < Nq;
    KpPair pr = kp_pair_fetch(kp_pair_index(idrow, pair_lane, cl, K, Ns), Ns, s, rowpos);  // chunk 0's pair
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int id_next = kp_pair_index(idrow, pair_lane, k0 + KC + cl, K, Ns);           // in flight during phase A
        {   // ---- phase A: thread = (query ql, neighbour k0 + cl) ----
            float w[KP_MAXP];
            const bool positive = kp_pair_influences<true>(P, pr, qx, qy, qz, w);
            kp_count_positive<LQ>(positive, ql, cl, lcnt);
            if (cl < KC) {
                lidx[ql * KC + cl] = pr.id;
                kp_store_w(&lw[ql * WS + cl * 16], cl, w);
            }
        }
        __syncthreads();
        pr = kp_pair_fetch(id_next, Ns, s, rowpos);                                         // in flight during phase B
        // ---- phase B: thread = (query ql, channels 4*cl .. 4*cl+3); PF feature rows requested before any is consumed ----
        const int kend = min(KC, K - k0);
        for (int kg = 0; kg < kend; kg += PF) {
            float4 fv[PF];
            int ids[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ids[u] = (kg + u < kend) ? lidx[ql * KC + kg + u] : -1;
                fv[u] = kp_gather4(fbuf, ids[u], ldf, 4 * cl);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (!__any(ids[u] >= 0)) continue;   // (wavefront-uniform) shadow slot for every query of the wavefront: nothing to add
                float w[16];
                kp_load_w(&lw[ql * WS + (kg + u) * 16], kg + u, w);
#pragma unroll
                for (int p = 0; p < KP_MAXP - 1; ++p) {
                    acc[p][0] = fmaf(w[p], fv[u].x, acc[p][0]);
                    acc[p][1] = fmaf(w[p], fv[u].y, acc[p][1]);
                    acc[p][2] = fmaf(w[p], fv[u].z, acc[p][2]);
                    acc[p][3] = fmaf(w[p], fv[u].w, acc[p][3]);
                }
            }
        }
        __syncthreads();
    }
    // ---- contraction: out[16 x Cout] = wf[16 x 15*Cin] @ K_values, the wf tile through LDS in passes of HP kernel points ----
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    kp_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
        // operand-split form: passes of 256 k-values (HP3 kernel points), the tile as three bf16 planes; products of one 32-deep
        // step: (wf plane, W plane) = (2,0) (1,0) (0,0) (1,1) (0,1) (0,2) on three accumulator chains by magnitude class
        constexpr int HP3 = KX_KT / CIN, NW = COUT / 16, PL3 = KG_TQ * KX_TS;
        static_assert(HP3 >= 1 && 3 * PL3 * 2 <= KG_TQ * KG_TS * 4, "the plane tile must fit the region");
        unsigned short* tile3 = (unsigned short*)region;                       // [3][16][KX_TS]
        kp_f32x4 cS = {0.f, 0.f, 0.f, 0.f};
        const int nsteps = P.num_kp * CIN / 32;                                // 32-deep steps over the whole k range (even)
        const uint4* bw = (const uint4*)Wp + (size_t)wave * 64 + lane;         // + ((step * 3 + plane) * NW) * 64
        uint4 b0[3], b1[3];
#define KX_BLOAD(B_, SG_)                                                                                          \
        do {                                                                                                       \
            const int sg_ = min((SG_), nsteps - 1);                                                                \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) B_[pl] = bw[(size_t)((sg_ * 3 + pl) * NW) * 64];      \
        } while (0)
        KX_BLOAD(b0, 0);
        KX_BLOAD(b1, 1);
        const unsigned short* ap = tile3 + r16 * KX_TS + 8 * g;
#define KX_STEP(B_, T_)                                                                                                        \
        do {                                                                                                                   \
            uint4 a_[3];                                                                                                       \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) a_[pl] = *(const uint4*)(ap + pl * PL3 + 32 * (T_));              \
            cS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[2]), __builtin_bit_cast(kp_bf16x8, B_[0]), cS, 0, 0, 0); \
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[0]), c1, 0, 0, 0); \
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[0]), c0, 0, 0, 0); \
            cS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[1]), __builtin_bit_cast(kp_bf16x8, B_[1]), cS, 0, 0, 0); \
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[1]), c1, 0, 0, 0); \
            cS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kp_bf16x8, a_[0]), __builtin_bit_cast(kp_bf16x8, B_[2]), cS, 0, 0, 0); \
        } while (0)
        const int npass3 = (P.num_kp + HP3 - 1) / HP3;
        int sg = 0;
        for (int pass = 0; pass < npass3; ++pass) {
            const int p0 = pass * HP3;
            const int np = min(P.num_kp - p0, HP3);
            if (pass) __syncthreads();                  // the previous pass's planes have been consumed
#pragma unroll
            for (int p = 0; p < KP_MAXP - 1; ++p) {
                const int pp = p - p0;
                if (pp >= 0 && pp < np) {               // (workgroup-uniform)
                    uint2 pl[3];
                    kx_split4(acc[p][0], acc[p][1], acc[p][2], acc[p][3], pl);
                    unsigned short* d = tile3 + ql * KX_TS + pp * CIN + 4 * cl;
                    *(uint2*)d = pl[0];
                    *(uint2*)(d + PL3) = pl[1];
                    *(uint2*)(d + 2 * PL3) = pl[2];
                }
            }
            __syncthreads();
            const int T = np * CIN / 32;                // (even: Cin is a multiple of 64)
            for (int t = 0; t < T; t += 2) {
                KX_STEP(b0, t);
                KX_BLOAD(b0, sg + 2);
                KX_STEP(b1, t + 1);
                KX_BLOAD(b1, sg + 3);
                sg += 2;
            }
        }
#undef KX_STEP
#undef KX_BLOAD
#pragma unroll
        for (int i = 0; i < 4; ++i) c1[i] += cS[i];     // small terms first, then the leading products (c0) in the epilogue's sum
    }
    const int npass = X3 ? 0 : (P.num_kp + HP - 1) / HP;
    for (int pass = 0; pass < npass; ++pass) {
        const int p0 = pass * HP;
        const int np = min(P.num_kp - p0, HP);
        if (pass) __syncthreads();                      // the previous pass's tile has been consumed
        // this thread's weighted features of the pass -> tile row ql, k = pp*Cin + 4*cl + {0..3}  (k order of K_values)
#pragma unroll
        for (int p = 0; p < KP_MAXP - 1; ++p) {
            const int pp = p - p0;
            if (pp >= 0 && pp < np)
                *(float4*)&wft[ql * KG_TS + pp * CIN + 4 * cl] = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        }
        __syncthreads();
        const int ngrp = np * CIN / 64;                 // groups of four 16-deep k-blocks (np*Cin is a multiple of 64)
        // B: packed K_values, float4 index ((blk*4 + g) * COUT + n); blocks of this pass start at p0*CIN/16.  Every address
        // is (wave-uniform block base, in scalar registers) + ONE 32-bit per-lane offset: a single address VGPR for all loads
        const float4* bpass = (const float4*)Wp + (size_t)(p0 * CIN / 16) * 4 * COUT;
        const unsigned boff = (unsigned)(g * COUT + 16 * wave + r16);
        const float* ap = &wft[r16 * KG_TS + 4 * g];
        constexpr unsigned BSTEP = 4u * COUT;           // float4s per k-block
#define KG_BLOAD(GRP_, U_) (bpass + (size_t)((unsigned)(4 * (GRP_) + (U_)) * BSTEP))[boff]
        float4 b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b0[u] = KG_BLOAD(0, u);
        {
            const int g1 = min(1, ngrp - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) b1[u] = KG_BLOAD(g1, u);
        }
#define KG_CONSUME(BQ_, GI_)                                                                     \
        do {                                                                                     \
            float4 a_ = *(const float4*)&ap[16 * (4 * (GI_))];                                   \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                      \
                const float4 an_ = *(const float4*)&ap[16 * (4 * (GI_) + (u < 3 ? u + 1 : u))];  \
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, BQ_[u].x, c0, 0, 0, 0);          \
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, BQ_[u].y, c1, 0, 0, 0);          \
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, BQ_[u].z, c0, 0, 0, 0);          \
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, BQ_[u].w, c1, 0, 0, 0);          \
                a_ = an_;                                                                        \
            }                                                                                    \
        } while (0)
        int gi = 0;
        for (; gi + 1 < ngrp; gi += 2) {
            // a buffer is refilled (group gi+2 / gi+3, clamped: straight-line code, no branch between a load and its use)
            // right after its multiplies are issued: every load has the other buffer's 16 multiplies (512 cycles) of cover
            const int ga = min(gi + 2, ngrp - 1), gb = min(gi + 3, ngrp - 1);
            KG_CONSUME(b0, gi);
#pragma unroll
            for (int u = 0; u < 4; ++u) b0[u] = KG_BLOAD(ga, u);
            KG_CONSUME(b1, gi + 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) b1[u] = KG_BLOAD(gb, u);
        }
        if (gi < ngrp) KG_CONSUME(b0, gi);              // odd group count: the last group sits in b0
#undef KG_CONSUME
#undef KG_BLOAD
    }
    // ---- epilogue in the accumulator registers: C/D layout of 16x16x4: col = lane & 15, row = 4*(lane >> 4) + i ----
    const int n = 16 * wave + r16;
    const float cs = E.col_scale ? E.col_scale[n] : 1.f;
    const float ch = E.col_shift ? E.col_shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * g + i;
        const int gq = lq[row];
        if (gq < 0) continue;
        float v = (c0[i] + c1[i]) * (1.0f / fmaxf((float)lcnt[row], 1.0f));
        v = v * cs + ch;
        if (E.residual) v += E.residual[(size_t)gq * E.ldr + n];
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
        D3fFeat<FT>::st1(&out[(size_t)gq * ldo + n], v);
    }
}

// Wp[blk][g][n][j] = W[16*blk + 4*g + j][n]   (K % 16 == 0): the B-operand order of kpconv_fused_kernel
__global__ void __launch_bounds__(256) kp_pack_weights_kernel(const float* __restrict__ W, int K, int N, float* __restrict__ Wp) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)K * N) return;
    const int j = (int)(t & 3);
    const long long u = t >> 2;
    const int n = (int)(u % N);
    const long long v = u / N;
    const int g = (int)(v & 3);
    const long long blk = v >> 2;
    Wp[t] = W[(size_t)(16 * blk + 4 * g + j) * N + n];
}

extern "C" int d3f_kpconv_pack_weights(const float* W, int K, int N, float* Wp, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 16 || (K % 16) || N < 1 || !W || !Wp) return D3F_ERR_ARG;
    kp_pack_weights_kernel<<<d3f_cdiv((long long)K * N, 256), 256, 0, stream>>>(W, K, N, Wp);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// Wx[((step * 3 + plane) * NW + w) * 64 + lane][j] = plane(W[32 step + 8 (lane >> 4) + j][16 w + (lane & 15)])   (K % 32 == 0,
// N % 16 == 0): the B fragments of v_mfma_f32_16x16x32_bf16 for the wave that owns output columns 16 w .. 16 w + 15
__global__ void __launch_bounds__(256) kp_pack_weights_x3_kernel(const float* __restrict__ W, int K, int N, unsigned short* __restrict__ Wx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3ll * K * N) return;
    const int j = (int)(t & 7);
    long long u = t >> 3;
    const int lane = (int)(u & 63); u >>= 6;
    const int NW = N / 16;
    const int w = (int)(u % NW); u /= NW;
    const int pl = (int)(u % 3);
    const int step = (int)(u / 3);
    float x = W[(size_t)(32 * step + 8 * (lane >> 4) + j) * N + 16 * w + (lane & 15)];
    unsigned h = d3f_bf16_rne(x);
    if (pl > 0) { x -= __uint_as_float(h << 16); h = d3f_bf16_rne(x); }
    if (pl > 1) { x -= __uint_as_float(h << 16); h = d3f_bf16_rne(x); }
    Wx[t] = (unsigned short)h;
}

// N = 32 (kpconv_fused32_kernel, v_mfma_f32_32x32x16_bf16): Wx[(step * 3 + plane) * 64 + lane][j] =
// plane(W[16 step + 8 (lane >> 5) + j][lane & 31])   (K % 16 == 0)
__global__ void __launch_bounds__(256) kp_pack_weights_x3_n32_kernel(const float* __restrict__ W, int K, unsigned short* __restrict__ Wx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3ll * K * 32) return;
    const int j = (int)(t & 7);
    long long u = t >> 3;
    const int lane = (int)(u & 63); u >>= 6;
    const int pl = (int)(u % 3);
    const int step = (int)(u / 3);
    float x = W[(size_t)(16 * step + 8 * (lane >> 5) + j) * 32 + (lane & 31)];
    unsigned h = d3f_bf16_rne(x);
    if (pl > 0) { x -= __uint_as_float(h << 16); h = d3f_bf16_rne(x); }
    if (pl > 1) { x -= __uint_as_float(h << 16); h = d3f_bf16_rne(x); }
    Wx[t] = (unsigned short)h;
}

extern "C" size_t d3f_kpconv_packed_x3_bytes(int K, int N) { return (K > 0 && N > 0) ? (size_t)K * (size_t)N * 6u : 0; }

extern "C" int d3f_kpconv_pack_weights_x3(const float* W, int K, int N, void* Wx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 32 || (K % 32) || N < 16 || (N % 16) || !W || !Wx) return D3F_ERR_ARG;
    if (N == 32)        // the level-0 kernel multiplies 32 x 32 tiles: its own fragment order
        kp_pack_weights_x3_n32_kernel<<<d3f_cdiv(3ll * K * 32, 256), 256, 0, stream>>>(W, K, (unsigned short*)Wx);
    else
    kp_pack_weights_x3_kernel<<<d3f_cdiv(3ll * K * N, 256), 256, 0, stream>>>(W, K, N, (unsigned short*)Wx);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// the fused kernel exists for the configuration of the shipped models only (kp_influences_t<true>); anything else takes the
// two-kernel form (d3f_kpconv_aggregate + d3f_gemm_f32)
// 1: the one-kernel form exists AND is the faster choice; 2: it exists (d3f_kpconv_fused accepts the shape) but the two-kernel
// form (aggregation + contraction) measured the same or better end to end -- Cin = 256: 1024-thread workgroups, 1471 / 1452 against
// 1457 / 1472 fragments/s, 1325 against 1351 on the 20-fragment job (profiles/r03_experiments.txt x7); 0: not available.
extern "C" int d3f_kpconv_fused_supported(int Cin, int Cout, int num_kp, int influence, int aggregation) {
    if (!(Cin == Cout && kp_fast_config(num_kp, influence, aggregation))) return 0;
    return (Cin == 64 || Cin == 128) ? 1 : (Cin == 256 ? 2 : 0);
}

template <bool X3>
static int kp_fused_launch(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                           const void* f_, int ldf, int Cin, const unsigned char* rowpos, const float* kp_host, int num_kp,
                           float KP_extent, int influence, int aggregation, const float* W_packed, int Cout,
                           const float* col_scale, const float* col_shift, const float* residual, int ldr, int leaky,
                           float alpha, void* out_, int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order,
                           int feat_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* f = (const float*)f_;
    float* out = (float*)out_;
    if (feat_bf16 && residual) return D3F_ERR_ARG;
    if (!d3f_kpconv_fused_supported(Cin, Cout, num_kp, influence, aggregation)) return D3F_ERR_ARG;
    if (Nq < 0 || Ns < 0 || K < 0 || ld_idx < K || ldf < Cin || (ldf % 4) || num_kp < 1 || num_kp > KP_MAXP - 1 ||
        influence < 0 || influence > 2 || aggregation < 0 || aggregation > 1 || !(KP_extent > 0.f) || ldo < Cout ||
        (residual && ldr < Cout))
        return D3F_ERR_ARG;
    if (Nq == 0) return D3F_OK;
    if (!q || !s || !idx || !f || !rowpos || !kp_host || !W_packed || !out || (((uintptr_t)f | (uintptr_t)W_packed) & 15))
        return D3F_ERR_ARG;
    if (!kp_fits_u24(Nq, Ns, ld_idx, ldf)) return D3F_ERR_ARG;     // (16.7 M rows per call: use d3f_kpconv_aggregate + d3f_gemm_f32)
    const KpParams P = kp_make_params(kp_host, num_kp, KP_extent, influence, aggregation);
    KpEpi E{col_scale, col_shift, residual, ldr, leaky, alpha};
    const int blocks = d3f_cdiv(Nq, KG_TQ);
#define D3F_KG(LQ_, PF_)                                                                                                    \
    kpconv_fused_kernel<LQ_, PF_, float, X3><<<blocks, KG_TQ * LQ_, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, rowpos, P, W_packed, \
                                                                                 E, out, ldo, Nq_dev, Ns_dev, q_order)
    if (feat_bf16) {
        const unsigned short* fh = (const unsigned short*)f_;
        unsigned short* oh = (unsigned short*)out_;
        if (Cin == 64)
            kpconv_fused_kernel<16, D3F_KP_PF_H, unsigned short, X3><<<blocks, KG_TQ * 16, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, fh, ldf, rowpos, P,
                                                                                         W_packed, E, oh, ldo, Nq_dev, Ns_dev, q_order);
        else if (Cin == 256)
            kpconv_fused_kernel<64, 4, unsigned short, X3><<<blocks, KG_TQ * 64, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, fh, ldf, rowpos, P,
                                                                                         W_packed, E, oh, ldo, Nq_dev, Ns_dev, q_order);
        else
            kpconv_fused_kernel<32, 4, unsigned short, X3><<<blocks, KG_TQ * 32, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, fh, ldf, rowpos, P,
                                                                                         W_packed, E, oh, ldo, Nq_dev, Ns_dev, q_order);
    } else if (Cin == 64) D3F_KG(16, D3F_KP_PF);
    else if (Cin == 256) D3F_KG(64, 4);
    else D3F_KG(32, 4);
#undef D3F_KG
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

#define KP_FUSED_ARGS                                                                                                             \
    const float *q, int Nq, const float *s, int Ns, const int *idx, int ld_idx, int K, const void *f_, int ldf, int Cin,          \
        const unsigned char *rowpos, const float *kp_host, int num_kp, float KP_extent, int influence, int aggregation,           \
        const float *W_packed, int Cout, const float *col_scale, const float *col_shift, const float *residual, int ldr,          \
        int leaky, float alpha, void *out_, int ldo, const int *Nq_dev, const int *Ns_dev, const int *q_order, int feat_bf16,     \
        void *stream_
#define KP_FUSED_PASS                                                                                                             \
    q, Nq, s, Ns, idx, ld_idx, K, f_, ldf, Cin, rowpos, kp_host, num_kp, KP_extent, influence, aggregation, W_packed, Cout,       \
        col_scale, col_shift, residual, ldr, leaky, alpha, out_, ldo, Nq_dev, Ns_dev, q_order, feat_bf16, stream_
extern "C" int d3f_kpconv_fused(KP_FUSED_ARGS) { return kp_fused_launch<false>(KP_FUSED_PASS); }
// the same operator with the contraction in the operand-split form; W_packed = d3f_kpconv_pack_weights_x3's planes
extern "C" int d3f_kpconv_fused_x3(KP_FUSED_ARGS) { return kp_fused_launch<true>(KP_FUSED_PASS); }
#undef KP_FUSED_ARGS
#undef KP_FUSED_PASS

// ---- C ABI ---------------------------------------------------------------------------------------
extern "C" int d3f_row_positive(const void* f_, int Ns, int ldf, int Cin, unsigned char* row_pos, const int* Ns_dev,
                                int feat_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* f = (const float*)f_;
    if (Ns < 0 || Cin < 1 || ldf < Cin) return D3F_ERR_ARG;
    if (Ns == 0) return D3F_OK;
    if (!f || !row_pos) return D3F_ERR_ARG;
    if (feat_bf16) {
        if (Cin % 4 || ldf % 4 || ((uintptr_t)f_ & 7)) return D3F_ERR_ARG;
        const unsigned short* fh = (const unsigned short*)f_;
        const int q4 = Cin / 4;
#define D3F_ROWPOS_H(LPR_) kp_rowpos_vec_kernel<LPR_, unsigned short><<<d3f_cdiv((long long)Ns * LPR_, 256), 256, 0, stream>>>(fh, Ns, Ns_dev, ldf, Cin, row_pos)
        if (q4 <= 8) D3F_ROWPOS_H(8);
        else if (q4 <= 16) D3F_ROWPOS_H(16);
        else if (q4 <= 32) D3F_ROWPOS_H(32);
        else D3F_ROWPOS_H(64);
#undef D3F_ROWPOS_H
        D3F_LAUNCH_CHECK();
        return D3F_OK;
    }
    if (Cin % 4 == 0 && ldf % 4 == 0 && ((uintptr_t)f & 15) == 0) {
        const int q4 = Cin / 4;
#define D3F_ROWPOS(LPR_) kp_rowpos_vec_kernel<LPR_><<<d3f_cdiv((long long)Ns * LPR_, 256), 256, 0, stream>>>(f, Ns, Ns_dev, ldf, Cin, row_pos)
        if (q4 <= 4) D3F_ROWPOS(4);
        else if (q4 <= 8) D3F_ROWPOS(8);
        else if (q4 <= 16) D3F_ROWPOS(16);
        else if (q4 <= 32) D3F_ROWPOS(32);
        else D3F_ROWPOS(64);
#undef D3F_ROWPOS
    } else {
        kp_rowpos_kernel<<<d3f_cdiv((long long)Ns * 64, 256), 256, 0, stream>>>(f, Ns, Ns_dev, ldf, Cin, row_pos);
    }
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" int d3f_kpconv_aggregate(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                                    const void* f_, int ldf, int Cin, const unsigned char* rowpos, const float* kp_host,
                                    int num_kp, float KP_extent, int influence, int aggregation, float* wf, float* inv_cnt,
                                    const int* Nq_dev, const int* Ns_dev, const int* q_order, int feat_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* f = (const float*)f_;
    if (Nq < 0 || Ns < 0 || K < 0 || ld_idx < K || Cin < 1 || ldf < Cin || num_kp < 1 || num_kp > KP_MAXP - 1 ||
        influence < 0 || influence > 2 || aggregation < 0 || aggregation > 1 || !(KP_extent > 0.f))
        return D3F_ERR_ARG;
    if (Nq == 0) return D3F_OK;
    if (!q || !s || !idx || !f || !rowpos || !kp_host || !wf || !inv_cnt) return D3F_ERR_ARG;
    const KpParams P = kp_make_params(kp_host, num_kp, KP_extent, influence, aggregation);
    // (the vector kernels address rows with 24-bit multiplies; larger problems take the one-thread-per-output kernel)
    const bool vec = (Cin % 4 == 0) && (ldf % 4 == 0) && (((uintptr_t)f & 15) == 0) && (((uintptr_t)wf & 15) == 0) &&
                     kp_fits_u24(Nq, Ns, ld_idx, ldf);
    const bool fast = kp_fast_config(num_kp, influence, aggregation);
    if (feat_bf16) {
        if (!kp_fits_u24(Nq, Ns, ld_idx, ldf)) return D3F_ERR_ARG;     // bf16 feature rows in, fp32 weighted features out; the deep layers of the shipped configuration
        const unsigned short* fh = (const unsigned short*)f_;
        if (!fast || (ldf % 4) || ((uintptr_t)f_ & 7) || ((uintptr_t)wf & 15) || !(Cin == 256 || Cin == 512)) return D3F_ERR_ARG;
        if (Cin == 256)
            kpconv_agg_vec4<64, true, unsigned short><<<d3f_cdiv(Nq, 4), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, fh, ldf, rowpos, P, wf,
                                                                                         inv_cnt, Nq_dev, Ns_dev, q_order);
        else
            kpconv_agg_vec4<128, true, unsigned short><<<d3f_cdiv(Nq, 2), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, fh, ldf, rowpos, P, wf,
                                                                                          inv_cnt, Nq_dev, Ns_dev, q_order);
        D3F_LAUNCH_CHECK();
        return D3F_OK;
    }
#define D3F_AGG(LQ_)                                                                                                        \
    do {                                                                                                                    \
        if (fast) kpconv_agg_vec4<LQ_, true><<<d3f_cdiv(Nq, 256 / LQ_), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf, \
                                                                                          rowpos, P, wf, inv_cnt, Nq_dev, Ns_dev, q_order); \
        else kpconv_agg_vec4<LQ_, false><<<d3f_cdiv(Nq, 256 / LQ_), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf,   \
                                                                                       rowpos, P, wf, inv_cnt, Nq_dev, Ns_dev, q_order); \
    } while (0)
    if (vec && Cin == 4) D3F_AGG(1);
    else if (vec && Cin == 8) D3F_AGG(2);
    else if (vec && Cin == 16) D3F_AGG(4);
    else if (vec && Cin == 32) D3F_AGG(8);
    else if (vec && Cin == 64) D3F_AGG(16);
    else if (vec && Cin == 128) D3F_AGG(32);
    else if (vec && Cin == 256) D3F_AGG(64);
    else if (vec && Cin == 512) D3F_AGG(128);
    else if (vec && Cin == 1024) D3F_AGG(256);
    else
        kpconv_agg_scalar<<<d3f_cdiv((long long)Nq * Cin, 256), 256, 0, stream>>>(q, Nq, s, Ns, idx, ld_idx, K, f, ldf,
                                                                                   Cin, rowpos, P, wf, inv_cnt, Nq_dev, Ns_dev);
#undef D3F_AGG
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
