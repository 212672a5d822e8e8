// One-workgroup form of the grid subsampling for SMALL clouds (included by grid_subsample.hip, capacity mode only).
//
// The coarse levels of the pyramid subsample a few thousand points per cloud (3DMatch: 8.2 k -> 2.2 k -> 650 -> 195).  As a
// chain of grid-wide kernels that is ~20 dependent launches per level, each a few microseconds of launch + drain for a few
// microseconds of work on a handful of CUs.  Here ONE workgroup per cloud runs the whole algorithm out of LDS:
//
//   bounding box -> origin / grid (grid_subsampling.cpp:24-30) -> voxel keys (:49-59) -> stable LSD radix sort of (key, point)
//   in LDS (wave-ballot ranking as in radix_sort.h; the tile is the whole cloud) -> run heads -> first-occurrence bits +
//   popcount scan = voxel ids in insertion order -> the head of a run sums its points in input order (:63-70, :81-92) ->
//   libstdc++ iteration-order rounds (closed form of grid_subsample.hip's header) on the voxel keys, all in LDS ->
//   barycentres to the cloud's block of a staging buffer [B][elem_cap][3];
//
// a second, trivial launch packs the B blocks into the contiguous [M, 3] output and writes lengths / status (the offsets need
// every cloud's voxel count).  Same results, bit for bit, as the other two forms (tests/gs_sort_path_check.py).
// Limits, checked on the device and reported like every other capacity (D3F_ST_OUT_OVERFLOW -> empty result, the caller's eager
// path recomputes): a cloud of more than T * R points, more than `nbmax` voxels in a cloud, a grid of more than 2^32 cells.
#pragma once

#define GSS_NB_MAX 5087

struct GsSmallArgs {
    const float* pts;      // [N_cap, 3]
    const int* lens;       // [B] device
    int B;
    float dl;
    float* stage;          // [B][elem_cap][3]   barycentres by iteration-order position
    float* recs;           // [B][elem_cap][4]   scratch: {barycentre, key} by voxel id (16-byte aligned)
    int elem_cap;
    int out_cap;           // rows of the final output
    int nbmax;             // largest bucket count whose round fits the LDS of this launch (a chain value)
    int* mcount;           // [B] voxels per cloud (0 when the cloud could not be processed)
    int* cflags;           // [B] D3F_ST_* bits of each cloud (written, not accumulated: nothing to reset between calls)
};

template <int T>
__device__ __forceinline__ int gss_block_excl_scan(int v, int* __restrict__ wsum, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < T / 64; ++q) {
        const int s = wsum[q];
        if (q < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// LDS layout (dynamic), in 32-bit words:
//   [0, 128)                   scratch: wave sums, box reduction; word 127 = the cloud's status bits
//   sort phase:   sK[T*R] sV[T*R] sCnt[T/64][256] sBits[T*R/32] sWS[T*R/32]
//   order phase:  (aliases the sort phase)  KEY[nbmax] BF BC BH NX L0 L1   (7 arrays of nbmax words)
template <int T, int R>
static inline size_t gss_lds_bytes(int nbmax) {
    const size_t sortw = 2 * (size_t)T * R + (size_t)(T / 64) * 256 + 2 * ((size_t)T * R / 32);
    const size_t orderw = 7 * (size_t)nbmax;
    return (128 + (sortw > orderw ? sortw : orderw)) * 4;
}

template <int T, int R>
__global__ void __launch_bounds__(T) gs_small_kernel(GsSmallArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned gss_lds[];
    constexpr int W = T / 64, NI = T * R;
    int* sScr = (int*)gss_lds;
    unsigned* sK = gss_lds + 128;
    unsigned* sV = sK + NI;
    unsigned (*sCnt)[256] = (unsigned (*)[256])(sV + NI);
    unsigned* sBits = sV + NI + W * 256;
    int* sWS = (int*)(sBits + NI / 32);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int lo = 0;
    for (int j = 0; j < b; ++j) lo += A.lens[j];
    const int len = A.lens[b];
    int* sFlag = sScr + 127;
    if (tid == 0) { A.mcount[b] = 0; *sFlag = 0; }
    if (len <= 0 || len > NI) {     // (an empty cloud is UB in the reference, cloud.cpp:30,51; a cloud beyond T * R points: capacity)
        if (tid == 0) A.cflags[b] = len <= 0 ? D3F_ST_EMPTY_ELEMENT : D3F_ST_OUT_OVERFLOW;
        return;
    }
    const float* __restrict__ P = A.pts + 3 * (size_t)lo;
    const float dl = A.dl;

    // ---- bounding box, origin, grid dimensions ------------------------------------------------------------------------
    unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (int i = tid; i < len; i += T) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned u = d3f_f2ord(P[3 * (size_t)i + d]);
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
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { sScr[w * 6 + d] = (int)mn[d]; sScr[w * 6 + 3 + d] = (int)mx[d]; }
    }
    __syncthreads();
    float org[3];
    unsigned dims[3];
    {
        const float inv = __fdiv_rn(1.0f, dl);  // `1/sampleDl`
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned a = 0xFFFFFFFFu, c = 0u;
            for (int q = 0; q < W; ++q) { a = min(a, (unsigned)sScr[q * 6 + d]); c = max(c, (unsigned)sScr[q * 6 + 3 + d]); }
            org[d] = __fmul_rn(floorf(__fmul_rn(d3f_ord2f(a), inv)), dl);
            const float ext = fmaxf(floorf(__fdiv_rn(__fsub_rn(d3f_ord2f(c), org[d]), dl)), 0.f) + 1.0f;
            dims[d] = ext < 4294967040.f ? (unsigned)ext : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    const double cells = (double)dims[0] * (double)dims[1] * (double)dims[2];
    if (cells > 4294967296.0) {
        if (tid == 0) A.cflags[b] = D3F_ST_KEY_WIDTH;
        return;
    }
    int kb = 0;
    {
        unsigned long long c1 = (unsigned long long)cells - 1ull;
        while (c1) { ++kb; c1 >>= 1; }
        if (kb < 1) kb = 1;
    }
    const int npass = (kb + 7) / 8;

    // ---- voxel keys; a wave owns 64 R consecutive points ----------------------------------------------------------------
    const int base = w * (64 * R) + lane;
    unsigned key[R], val[R];
    int st = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = base + r * 64;
        key[r] = 0u;
        val[r] = (unsigned)i;
        if (i < len) {
            const float fx = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 0], org[0]), dl));
            const float fy = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 1], org[1]), dl));
            const float fz = floorf(__fdiv_rn(__fsub_rn(P[3 * (size_t)i + 2], org[2]), dl));
            if (fx < 0.f || fy < 0.f || fz < 0.f) st |= D3F_ST_NEG_CELL;
            const unsigned ix = (unsigned)fmaxf(fx, 0.f), iy = (unsigned)fmaxf(fy, 0.f), iz = (unsigned)fmaxf(fz, 0.f);
            key[r] = ix + dims[0] * (iy + dims[1] * iz);       // < cells <= 2^32
        }
    }
    if (st) atomicOr(sFlag, st);

    // ---- stable LSD radix sort of (key, point) in LDS ---------------------------------------------------------------------
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * 8;
        for (int t = tid; t < W * 256; t += T) (&sCnt[0][0])[t] = 0u;
        __syncthreads();
        unsigned rank[R];
        volatile unsigned* myCnt = sCnt[w];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = (base + r * 64) < len;
            const unsigned dg = (key[r] >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (dg >> bit) & 1u;
                const unsigned long long m = __ballot(one);
                peers &= one ? m : ~m;
            }
            rank[r] = 0u;
            if (valid) {
                const unsigned prev = myCnt[dg];
                const unsigned below = (unsigned)__popcll(peers & d3f_lanemask_lt());
                rank[r] = prev + below;
                __builtin_amdgcn_wave_barrier();
                if (below == 0u) myCnt[dg] = prev + (unsigned)__popcll(peers);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        {   // digit totals -> first position of each digit, then each wave's base inside the digit's run
            int cnt = 0;
            if (tid < 256)
                for (int q = 0; q < W; ++q) cnt += (int)sCnt[q][tid];
            int tot;
            const int loc = gss_block_excl_scan<T>(tid < 256 ? cnt : 0, sScr, &tot);
            if (tid < 256) {
                unsigned run = (unsigned)loc;
                for (int q = 0; q < W; ++q) {
                    const unsigned c = sCnt[q][tid];
                    sCnt[q][tid] = run;
                    run += c;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((base + r * 64) < len) {
                const unsigned pl = myCnt[(key[r] >> shift) & 255u] + rank[r];
                sK[pl] = key[r];
                sV[pl] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = base + r * 64;
            if (i < len) { key[r] = sK[i]; val[r] = sV[i]; }
        }
        // (the next pass writes sK / sV only after two more barriers)
    }

    // ---- run heads -> first-occurrence bits -> voxel ids ------------------------------------------------------------------
    for (int t = tid; t < NI / 32; t += T) sBits[t] = 0u;
    __syncthreads();
    unsigned headmask = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = base + r * 64;
        if (i < len && (i == 0 || sK[i - 1] != key[r])) {
            headmask |= 1u << r;
            atomicOr(&sBits[val[r] >> 5], 1u << (val[r] & 31u));
        }
    }
    __syncthreads();
    int M;
    {
        int carry = 0;
        for (int c0 = 0; c0 < NI / 32; c0 += T) {       // (one trip unless R > 32)
            const int t = c0 + tid;
            const int c = (t < NI / 32) ? __popc(sBits[t]) : 0;
            int tot;
            const int e = gss_block_excl_scan<T>(c, sScr, &tot);
            if (t < NI / 32) sWS[t] = carry + e;
            carry += tot;
        }
        M = carry;
    }
    __syncthreads();
    if (M > A.nbmax || M > A.elem_cap) {
        if (tid == 0) A.cflags[b] = *sFlag | D3F_ST_OUT_OVERFLOW;
        return;
    }

    // ---- the head of a run: voxel id, key, in-order barycentre -> the cloud's scratch records [vid] = {x, y, z, key} ---------
    float4* __restrict__ rec = (float4*)A.recs + (size_t)b * (size_t)A.elem_cap;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (headmask & (1u << r)) {
            const int i = base + r * 64;
            const unsigned k = key[r];
            unsigned pi = val[r];
            const int vid = sWS[pi >> 5] + __popc(sBits[pi >> 5] & ((1u << (pi & 31u)) - 1u));
            float sx = P[3 * (size_t)pi + 0], sy = P[3 * (size_t)pi + 1], sz = P[3 * (size_t)pi + 2];
            int c = 1;
            while (i + c < len && sK[i + c] == k) {
                pi = sV[i + c];
                sx = __fadd_rn(sx, P[3 * (size_t)pi + 0]);
                sy = __fadd_rn(sy, P[3 * (size_t)pi + 1]);
                sz = __fadd_rn(sz, P[3 * (size_t)pi + 2]);
                ++c;
            }
            const float sc = (float)(1.0 / (double)c);
            rec[vid] = make_float4(__fmul_rn(sx, sc), __fmul_rn(sy, sc), __fmul_rn(sz, sc), __uint_as_float(k));
        }
    }
    __syncthreads();       // the sort buffers are dead from here on: the order arrays take their place (the records written
                           // above are read back by this same workgroup: visible after the barrier)

    // ---- libstdc++ iteration order of the M voxel keys, every round in LDS --------------------------------------------------
    const int NB = A.nbmax;
    unsigned* sKey = gss_lds + 128;
    int* sBF = (int*)(sKey + NB);
    int* sBC = sBF + NB;
    int* sBH = sBC + NB;
    int* sNX = sBH + NB;
    int* sL0 = sNX + NB;
    int* sL1 = sL0 + NB;
    for (int v = tid; v < M; v += T) sKey[v] = __float_as_uint(rec[v].w);
    __syncthreads();
    int* La = sL0;
    int* Lb = sL1;
    int lo_r = 0;
    for (int j = 0; j < D3F_NCHAIN; ++j) {
        const int nb = (int)D3F_CHAIN_DEV[j];
        const double inv_nb = 1.0 / (double)nb;
        const bool last = M <= nb;
        const int hi = last ? M : nb;
        for (int t = tid; t < nb; t += T) { sBF[t] = 0x7fffffff; sBC[t] = 0; sBH[t] = -1; }
        __syncthreads();
        for (int t = tid; t < hi; t += T) {
            const int id = (t < lo_r) ? La[t] : t;
            const int bk = gs_mod((unsigned long long)sKey[id], (unsigned)nb, inv_nb);
            atomicMin(&sBF[bk], t);
            atomicAdd(&sBC[bk], 1);
            sNX[t] = atomicExch(&sBH[bk], t);
        }
        __syncthreads();
        // reverse exclusive scan over positions of c[t] = (t first of its bucket) ? bucket size : 0; the value replaces the
        // bucket's size (only its first position reads it)
        int carry = 0;
        for (int c0 = 0; c0 < hi; c0 += T) {
            const int u = c0 + tid, t = hi - 1 - u;
            int c = 0, bk = 0;
            if (u < hi) {
                const int id = (t < lo_r) ? La[t] : t;
                bk = gs_mod((unsigned long long)sKey[id], (unsigned)nb, inv_nb);
                c = (sBF[bk] == t) ? sBC[bk] : 0;
            }
            int tot;
            const int e = gss_block_excl_scan<T>(c, sScr, &tot);
            if (u < hi && sBF[bk] == t) sBC[bk] = carry + e;
            carry += tot;
        }
        __syncthreads();
        for (int t = tid; t < hi; t += T) {
            const int id = (t < lo_r) ? La[t] : t;
            const int bk = gs_mod((unsigned long long)sKey[id], (unsigned)nb, inv_nb);
            int rk = 0;
            for (int q = sBH[bk]; q >= 0; q = sNX[q]) rk += (q > t) ? 1 : 0;
            const int dest = sBC[bk] + rk;
            if (last) Lb[id] = dest;      // final round: the list buffer becomes the position table
            else Lb[dest] = id;
        }
        __syncthreads();
        int* tmp = La; La = Lb; Lb = tmp;
        if (last) break;
        lo_r = hi;
    }
    // La[v] = iteration-order position of voxel v
    float* __restrict__ out = A.stage + 3 * (size_t)b * (size_t)A.elem_cap;
    for (int v = tid; v < M; v += T) {
        const float4 q = rec[v];
        const size_t d = (size_t)La[v];
        out[3 * d + 0] = q.x;
        out[3 * d + 1] = q.y;
        out[3 * d + 2] = q.z;
    }
    if (tid == 0) { A.mcount[b] = M; A.cflags[b] = *sFlag; }
}

// staging blocks -> contiguous rows; lengths and status (one workgroup per 256 rows; every thread recomputes the offsets)
__global__ void __launch_bounds__(256) gs_small_pack_kernel(GsSmallArgs A, float* __restrict__ out, int* __restrict__ sub_lens,
                                                            int* __restrict__ status_dev) {
    __shared__ int sOff[D3F_MAX_BATCH + 1];
    if (threadIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < A.B; ++b) { sOff[b] = s; s += A.mcount[b]; }
        sOff[A.B] = s;
    }
    __syncthreads();
    const int M = sOff[A.B];
    int flags = 0;
    for (int b = 0; b < A.B; ++b) flags |= A.cflags[b];
    const bool over = M > A.out_cap || (flags & (D3F_ST_OUT_OVERFLOW | D3F_ST_KEY_WIDTH));
    if (blockIdx.x == 0) {
        for (int b = threadIdx.x; b < A.B; b += 256) sub_lens[b] = over ? 0 : A.mcount[b];
        if (threadIdx.x == 0) {
            const int fl = flags | ((M > A.out_cap) ? D3F_ST_OUT_OVERFLOW : 0);
            status_dev[0] = over ? 0 : M;
            status_dev[1] = fl;
        }
    }
    if (over) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    int b = 0;
    while (b + 1 < A.B && i >= sOff[b + 1]) ++b;
    const float* __restrict__ src = A.stage + 3 * ((size_t)b * (size_t)A.elem_cap + (size_t)(i - sOff[b]));
    out[3 * (size_t)i + 0] = src[0];
    out[3 * (size_t)i + 1] = src[1];
    out[3 * (size_t)i + 2] = src[2];
}
