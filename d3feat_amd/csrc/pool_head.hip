// Gather-type ops of the D3Feat network on gfx950: strided-shortcut max pooling, nearest upsampling (+ skip
// concatenation) and the descriptor / detection head.  All are row gathers: lanes map to channels so that each
// gathered row is one contiguous, coalesced read (16 B per lane where the channel count allows).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// ind_max_pool -- models/network_blocks.py:51-66.  x' = x with one extra row holding the per-channel minimum
// of x (the "shadow" row); out[n,c] = max_k x'[idx[n,k], c].
//
// The shadow row can only win where a row has NO valid neighbour at all (a column minimum is <= every real entry), which
// a pooled barycentre never is -- its own voxel's points lie inside the pooling radius.  So the full pass over x that the
// column minima cost (the whole feature matrix of the finer level, read only for them) is made lazy and stays inside the
// ONE pooling launch: a row's threads take the maximum over the valid neighbours, and the threads of a row without any
// stream the column minima of their own channels (the row's threads cover one whole row of x per step, coalesced).
// Result: bit-identical to the eager formulation in all cases; the rare path costs N1 row reads per such row.
// (Rounds 1-5 kept the minima in a scratch vector behind a flag: three more launches per pooling layer that left at once.)
// ------------------------------------------------------------------------------------------------
// U24: every row count / leading dimension fits the 24-bit addressing of common.h (d3f_fits_u24) -- the launcher decides
template <int VEC, class FT = float, bool U24 = false>  // channels per thread (4: 16-byte loads; 1: generic)
__global__ void __launch_bounds__(256) maxpool_kernel(const FT* __restrict__ x, int N1, int ldx, int C,
                                                      const int* __restrict__ idx, int N2, int ld_idx, int K,
                                                      FT* __restrict__ out, int ldo,
                                                      const int* __restrict__ N1_dev, const int* __restrict__ N2_dev,
                                                      const int* __restrict__ row_order) {
    N1 = d3f_dyn(N1, N1_dev);
    N2 = d3f_dyn(N2, N2_dev);
    const int CV = C / VEC;
    const unsigned nblk = (unsigned)(((long long)N2 * CV + 255) / 256);
    if (blockIdx.x >= nblk) return;                          // capacity-sized grid
    const long long t = (long long)d3f_xcd_tile(blockIdx.x, nblk) * 256 + threadIdx.x;   // one contiguous run of rows per XCD
    if (t >= (long long)N2 * CV) return;
    int slot, c;
    if (U24) { const unsigned tu = (unsigned)t; slot = (int)(tu / (unsigned)CV); c = (int)(tu - (unsigned)slot * (unsigned)CV) * VEC; }
    else { slot = (int)(t / CV); c = (int)(t % CV) * VEC; }
    const int n = row_order ? row_order[slot] : slot;   // spatially coherent visiting order (see kpconv.hip)
    float m[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) m[v] = -3.402823466e38f;
    const int* row = U24 ? idx + __umul24((unsigned)n, (unsigned)ld_idx) : idx + (size_t)n * ld_idx;
    int nvalid = 0;
    constexpr int NB = 8;        // neighbour rows requested per batch (the kernel is rounds x dependent round trips at full
                                 // occupancy with 30 registers: eight loads in flight still fit 8 waves per SIMD)
    for (int k0 = 0; k0 < K; k0 += NB) {
        int id[NB];
        float val[NB][VEC];
#pragma unroll
        for (int j = 0; j < NB; ++j) id[j] = (k0 + j < K) ? row[k0 + j] : -2;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool ok = id[j] >= 0 && id[j] < N1;
            nvalid += ok ? 1 : 0;
            if (ok) {
                const FT* xr = U24 ? x + (__umul24((unsigned)id[j], (unsigned)ldx) + (unsigned)c) : x + ((size_t)id[j] * ldx + c);
                if (VEC == 4) {
                    const float4 f = D3fFeat<FT>::ld4(xr);
                    val[j][0] = f.x; val[j][1 % VEC] = f.y; val[j][2 % VEC] = f.z; val[j][3 % VEC] = f.w;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) val[j][v] = D3fFeat<FT>::ld1(xr + v);
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) val[j][v] = -3.402823466e38f;  // shadow (<= every real entry) or beyond K
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], val[j][v]);
    }
    if (nvalid == 0) {   // this row is the shadow row: the column minima of x (ordered-uint minimum, as reduce_min keeps -0 < +0 apart)
        unsigned om[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) om[v] = 0xFFFFFFFFu;
        for (int r0 = 0; r0 < N1; r0 += 4) {
            float val[4][VEC];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = min(r0 + j, N1 - 1);
                const FT* xr = x + ((size_t)r * ldx + c);
                if (VEC == 4) {
                    const float4 f = D3fFeat<FT>::ld4(xr);
                    val[j][0] = f.x; val[j][1 % VEC] = f.y; val[j][2 % VEC] = f.z; val[j][3 % VEC] = f.w;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) val[j][v] = D3fFeat<FT>::ld1(xr + v);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) om[v] = min(om[v], d3f_f2ord(val[j][v]));
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v] = d3f_ord2f(om[v]);
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) D3fFeat<FT>::st1(&out[(size_t)n * ldo + c + v], m[v]);   // (a maximum of bf16 values is a bf16 value)
}

extern "C" int d3f_ind_max_pool(const void* x_, int N1, int ldx, int C, const int* idx, int N2, int ld_idx, int K,
                                void* out_, int ldo, float* col_min_dev, const int* N1_dev, const int* N2_dev,
                                const int* row_order, int feat_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* x = (const float*)x_;
    float* out = (float*)out_;
    (void)col_min_dev;   // scratch of rounds 1-5 (kept in the signature; may be NULL)
    if (N1 < 0 || N2 < 0 || C < 1 || ldx < C || ldo < C || K < 0 || ld_idx < K) return D3F_ERR_ARG;
    if (N2 == 0) return D3F_OK;
    if (!x || !idx || !out) return D3F_ERR_ARG;
    const bool u24 = d3f_fits_u24(N1, ldx) && d3f_fits_u24(N2, ld_idx) && (long long)N2 * C < (1ll << 31);
    if (feat_bf16) {
        if (C % 4 || ldx % 4 || ((uintptr_t)x_ & 7)) return D3F_ERR_ARG;
        const unsigned short* xh = (const unsigned short*)x_;
        unsigned short* oh = (unsigned short*)out_;
        if (u24)
            maxpool_kernel<4, unsigned short, true><<<d3f_cdiv((long long)N2 * (C / 4), 256), 256, 0, stream>>>(
                xh, N1, ldx, C, idx, N2, ld_idx, K, oh, ldo, N1_dev, N2_dev, row_order);
        else
            maxpool_kernel<4, unsigned short><<<d3f_cdiv((long long)N2 * (C / 4), 256), 256, 0, stream>>>(
                xh, N1, ldx, C, idx, N2, ld_idx, K, oh, ldo, N1_dev, N2_dev, row_order);
        D3F_LAUNCH_CHECK();
        return D3F_OK;
    }
    if (C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && u24)
        maxpool_kernel<4, float, true><<<d3f_cdiv((long long)N2 * (C / 4), 256), 256, 0, stream>>>(x, N1, ldx, C, idx, N2, ld_idx, K,
                                                                                                   out, ldo, N1_dev, N2_dev, row_order);
    else if (C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0)
        maxpool_kernel<4><<<d3f_cdiv((long long)N2 * (C / 4), 256), 256, 0, stream>>>(x, N1, ldx, C, idx, N2, ld_idx, K,
                                                                                      out, ldo, N1_dev, N2_dev, row_order);
    else
        maxpool_kernel<1><<<d3f_cdiv((long long)N2 * C, 256), 256, 0, stream>>>(x, N1, ldx, C, idx, N2, ld_idx, K,
                                                                               out, ldo, N1_dev, N2_dev, row_order);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// closest_pool (models/network_blocks.py:69-83, used by nearest_upsample_block :971-979) fused with the skip
// concatenation of models/D3Feat.py:63:  out[n] = [ x'[idx[n,0]] , skip[n] ],  x' = x + zero row.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) upsample_cat_kernel(const float* __restrict__ x, int N1, int ldx, int C1,
                                                           const int* __restrict__ idx, int N2, int ld_idx,
                                                           const float* __restrict__ skip, int lds, int C2,
                                                           float* __restrict__ out, int ldo, const int* __restrict__ N1_dev,
                                                           const int* __restrict__ N2_dev) {
    N1 = d3f_dyn(N1, N1_dev);
    N2 = d3f_dyn(N2, N2_dev);
    const int Ct = C1 + C2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N2 * Ct) return;
    const int n = (int)(t / Ct), c = (int)(t % Ct);
    float v;
    if (c < C1) {
        const int id = idx[(size_t)n * ld_idx];
        v = (id >= 0 && id < N1) ? x[(size_t)id * ldx + c] : 0.f;
    } else {
        v = skip[(size_t)n * lds + (c - C1)];
    }
    out[(size_t)n * ldo + c] = v;
}

extern "C" int d3f_closest_pool_cat(const float* x, int N1, int ldx, int C1, const int* idx, int N2, int ld_idx,
                                    const float* skip, int lds, int C2, float* out, int ldo, const int* N1_dev,
                                    const int* N2_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N1 < 0 || N2 < 0 || C1 < 1 || C2 < 0 || ldx < C1 || ldo < C1 + C2 || ld_idx < 1 || (C2 > 0 && lds < C2))
        return D3F_ERR_ARG;
    if (N2 == 0) return D3F_OK;
    if (!x || !idx || !out || (C2 > 0 && !skip)) return D3F_ERR_ARG;
    upsample_cat_kernel<<<d3f_cdiv((long long)N2 * (C1 + C2), 256), 256, 0, stream>>>(x, N1, ldx, C1, idx, N2, ld_idx, skip,
                                                                                      lds, C2, out, ldo, N1_dev, N2_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// D3Feat head -- models/D3Feat.py:65-115.
//   desc  = x * rsqrt(max(sum x^2, 1e-10))                                                   (:65)
//   m_b   = max over all entries of cloud b's rows (and 0 if the cloud's in_batches row is padded)   (:84-85)
//   y     = x / (m_b + 1e-6)                                                                 (:90)
//   mean  = sum_k y[idx[n,k]] / max(#{k : sum_c y[idx[n,k],c] != 0}, 1)                      (:93-97)
//   score = max_c softplus(y - mean) * y / (1e-6 + max_c y)                                  (:98-106)
// Kernel 1: per-cloud maxima (ordered-uint atomicMax).  Kernel 2: one half-wave (32 lanes) per point when
// C <= 32 (the shipped 32-d descriptor), generally ceil(C/32) channels per lane; each neighbour row is one
// coalesced 128-byte read; channel reductions are 5-step xor shuffles inside the 32-lane half.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_max_init_kernel(const int* __restrict__ lens, const int* __restrict__ include_zero_dev,
                                                            int group, int B, unsigned* __restrict__ mx, int* __restrict__ offs) {
    // include_zero_dev == NULL: derive it from the lengths (datasets/common.py:453-496: a row of in_batches holds the
    // shadow index iff the cloud is shorter than the longest one, or all clouds have the same length) -- inside every
    // group of `group` consecutive clouds (one reference stack each; group <= 0: the whole stack is one).
    // One workgroup, thread b = cloud b (B <= D3F_MAX_BATCH < 256): every length is ONE load into LDS, the rest reads LDS
    // (a serial loop over the clouds is a chain of dependent memory round trips on every replay's critical path).
    __shared__ int slen[256];
    const int b = threadIdx.x;
    const int len = b < B ? lens[b] : 0;
    const int inc_in = (include_zero_dev && b < B) ? include_zero_dev[b] : 0;
    slen[b] = len;
    __syncthreads();
    if (b >= B) return;
    const int g = group > 0 ? group : B;
    const int g0 = b / g * g, g1 = min(g0 + g, B);           // this cloud's group
    int before = 0;
    for (int j = 0; j < b; ++j) before += slen[j];
    int longest = 0, all_eq = 1;
    for (int j = g0; j < g1; ++j) longest = max(longest, slen[j]);
    for (int j = g0; j < g1; ++j) all_eq &= (slen[j] == longest);
    offs[b] = before;
    const int inc = include_zero_dev ? inc_in : ((len < longest || all_eq) ? 1 : 0);
    mx[b] = inc ? d3f_f2ord(0.f) : 0u;
    if (b == B - 1) offs[B] = before + len;
}

// per-cloud maximum of x (ordered-uint atomicMax).  DENSE: ldx == C and 16-byte aligned -> the cloud's rows are one flat
// float4 range (no index arithmetic per element)
template <bool DENSE>
__global__ void __launch_bounds__(256) head_max_kernel(const float* __restrict__ x, int N, int ldx, int C,
                                                       const int* __restrict__ offs, int B, unsigned* __restrict__ mx) {
    const int b = blockIdx.y;
    const long long lo = (long long)offs[b] * C, hi = (long long)offs[b + 1] * C;
    unsigned m = 0u;
    if (DENSE) {
        const long long lo4 = lo >> 2, hi4 = hi >> 2;        // C % 4 == 0, so both ends are float4 boundaries
        const long long stride = (long long)gridDim.x * blockDim.x;
        long long t = lo4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        for (; t + 3 * stride < hi4; t += 4 * stride) {      // four independent loads in flight per lane
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ((const float4*)x)[t + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = max(max(m, d3f_f2ord(v[u].x)), max(d3f_f2ord(v[u].y), max(d3f_f2ord(v[u].z), d3f_f2ord(v[u].w))));
        }
        for (; t < hi4; t += stride) {
            const float4 v = ((const float4*)x)[t];
            m = max(max(m, d3f_f2ord(v.x)), max(d3f_f2ord(v.y), max(d3f_f2ord(v.z), d3f_f2ord(v.w))));
        }
    } else {
        for (long long t = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; t < hi; t += (long long)gridDim.x * blockDim.x) {
            const long long r = t / C;
            const int c = (int)(t % C);
            m = max(m, d3f_f2ord(x[(size_t)r * ldx + c]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && lo < hi) atomicMax(&mx[b], max(max(red[0], red[1]), max(red[2], red[3])));
}

template <int CPL>  // channels per lane: C <= 32 * CPL
__global__ void __launch_bounds__(256)
head_kernel(const float* __restrict__ x, int N, int ldx, int C, const int* __restrict__ idx, int ld_idx, int K,
            const int* __restrict__ offs, int B, const unsigned* __restrict__ mx, float* __restrict__ desc, int ldd,
            float* __restrict__ score, const int* __restrict__ row_order) {
    N = min(N, offs[B]);   // N is the capacity, offs[B] the real point count
    if ((int)(blockIdx.x * 8) >= N) return;                  // capacity-sized grid (8 rows per workgroup)
    const int half = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((N + 7) / 8)) * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool active = half < N;
    const int n = active ? (row_order ? row_order[half] : half) : 0;   // spatially coherent visiting order
    const int b = d3f_find_elem(offs, B, n);
    const float den = d3f_ord2f(mx[b]) + 1e-6f;
    float xv[CPL], yv[CPL], sum[CPL];
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = l + 32 * j;
        xv[j] = (c < C) ? x[(size_t)n * ldx + c] : 0.f;
        yv[j] = xv[j] / den;
        sum[j] = 0.f;
        sq += xv[j] * xv[j];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    int cnt = 0;
    // the point's index row is fetched 32 entries at a time by the half-wave (coalesced), then broadcast entry by
    // entry; 4 neighbour rows are kept in flight
    const int hbase = threadIdx.x & 32;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int mine = (k0 + l < K) ? idx[(size_t)n * ld_idx + k0 + l] : -1;
        const int kn = min(32, K - k0);
        for (int kk = 0; kk < kn; kk += 4) {
            int id[4];
            float dn[4];
            float v[4][CPL];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                id[j] = __shfl(mine, hbase + min(kk + j, 31), 64);
                if (kk + j >= kn || id[j] < 0 || id[j] >= N) id[j] = -1;   // shadow row: zeros
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dn[j] = 1.f;
                if (id[j] >= 0) dn[j] = d3f_ord2f(mx[d3f_find_elem(offs, B, id[j])]) + 1e-6f;
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) {
                    const int c = l + 32 * jj;
                    v[j][jj] = (id[j] >= 0 && c < C) ? x[(size_t)id[j] * ldx + c] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float rs = 0.f;
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) {
                    const float y = v[j][jj] / dn[j];
                    sum[jj] += y;
                    rs += y;
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) rs += __shfl_xor(rs, o, 64);
                cnt += (id[j] >= 0 && rs != 0.f) ? 1 : 0;
            }
        }
    }
    const float fc = (float)max(cnt, 1);
    float ymax = -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
        if (l + 32 * j < C) ymax = fmaxf(ymax, yv[j]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
    float best = -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (l + 32 * j < C) {
            const float d = yv[j] - sum[j] / fc;
            // softplus as TF computes it: log1p(exp(d)) with the large/small-argument shortcuts
            float sp;
            if (d > 15.f) sp = d;
            else if (d < -15.f) sp = expf(d);
            else sp = log1pf(expf(d));
            best = fmaxf(best, sp * (yv[j] / (1e-6f + ymax)));
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (active) {
        const float inv = rsqrtf(fmaxf(sq, 1e-10f));
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            if (l + 32 * j < C) desc[(size_t)n * ldd + l + 32 * j] = xv[j] * inv;
        if (l == 0) score[n] = best;
    }
}

// C == 32 (the shipped descriptor width): 8 lanes x float4 per row, so one half-wave instruction fetches FOUR neighbour rows
// (4 row slots x 8 lanes) instead of one; the per-row sum for the non-zero count is a 3-step shuffle per four rows instead
// of a 5-step one per row.  Same arithmetic per element as head_kernel except y = v * (1 / den) (reciprocal multiply).
// "Does this row count as a neighbour": sum_c y[n, c] != 0 (models/D3Feat.py:94-96 counts the non-zero neighbour-feature sums).
// A property of the ROW, so it is evaluated once per row -- with exactly the summation tree head32_kernel used per (point,
// neighbour) pair: lane j of a row held channels 4j..4j+3, s_j = (y0 + y1) + (y2 + y3), then xor-shuffle sums over 1, 2, 4 --
// instead of 42 times per row inside the gather loop (three shuffles + adds + a compare per neighbour row and lane).
__global__ void __launch_bounds__(256) head32_rowflag_kernel(const float* __restrict__ x, int N, int ldx, const int* __restrict__ offs,
                                                             int B, const unsigned* __restrict__ mx, unsigned char* __restrict__ nz) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= min(N, offs[B])) return;
    const int b = d3f_find_elem(offs, B, n);
    const float rden = 1.0f / (d3f_ord2f(mx[b]) + 1e-6f);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = *(const float4*)&x[(size_t)n * ldx + 4 * j];
        s[j] = (v.x * rden + v.y * rden) + (v.z * rden + v.w * rden);
    }
    const float t0 = s[0] + s[1], t1 = s[2] + s[3], t2 = s[4] + s[5], t3 = s[6] + s[7];
    const float rs = (t0 + t1) + (t2 + t3);
    nz[n] = rs != 0.f ? 1 : 0;
}

template <bool U24>
__global__ void __launch_bounds__(256)
head32_kernel(const float* __restrict__ x, int N, int ldx, const int* __restrict__ idx, int ld_idx, int K,
              const int* __restrict__ offs, int B, const unsigned* __restrict__ mx, const unsigned char* __restrict__ nz,
              float* __restrict__ desc, int ldd, float* __restrict__ score, const int* __restrict__ row_order) {
    N = min(N, offs[B]);
    if ((int)(blockIdx.x * 8) >= N) return;                  // capacity-sized grid (8 rows per workgroup)
    const int half = (int)d3f_xcd_tile(blockIdx.x, (unsigned)((N + 7) / 8)) * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool active = half < N;
    const int n = active ? (row_order ? row_order[half] : half) : 0;
    const int slot = l >> 3, c4 = (l & 7) << 2;     // row slot 0..3, first of this lane's 4 channels
    const int b = d3f_find_elem(offs, B, n);
    const float den = d3f_ord2f(mx[b]) + 1e-6f;
    const float rden = 1.0f / den;
    const float4 xv = *(const float4*)&x[(size_t)n * ldx + c4];
    const float4 yv = make_float4(xv.x * rden, xv.y * rden, xv.z * rden, xv.w * rden);
    float sq = xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    // (accumulated as two register pairs: v_pk_mul / v_pk_add_f32 -- the same multiply and add per element, half the issue slots)
    typedef float hd_f2 __attribute__((ext_vector_type(2)));
    hd_f2 sum01 = {0.f, 0.f}, sum23 = {0.f, 0.f};
    int cnt = 0;
    const int hbase = threadIdx.x & 32;
    // the feature matrix as a buffer resource (U24 launches: rows * ldx * 4 bytes < 2^32)
    const __amdgpu_buffer_rsrc_t xbuf = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)N * (unsigned)ldx * 4u), 0x00020000);
    for (int k0 = 0; k0 < K; k0 += 32) {
        int mine = (k0 + l < K) ? idx[(size_t)n * ld_idx + k0 + l] : -1;   // 32 indices per half-wave load
        const bool valid = mine >= 0 && mine < N;                            // shadow row (or beyond K): zeros, never counted
        // neighbour count: the row flags of these 32 neighbours, one byte load per lane, one ballot per chunk
        const bool counts = valid && nz[valid ? mine : 0] != 0;
        cnt += __popcll((__ballot(counts) >> hbase) & 0xFFFFFFFFull);
        const int kn = min(32, K - k0);
        if (U24) {
            // buffer loads (24-bit row addressing: common.h).  A shadow slot names row N -- one past the buffer: the hardware's range
            // check returns exact zeros for it (never row 0's values times a zero factor: 0 * Inf = NaN), with no test per row; and
            // 0 * rden adds nothing, so the rows are accumulated without a select either (round 6: the kernel is vector-issue bound,
            // the loop went from 54 to ~30 vector instructions per 16 rows).  Slots beyond K hold row N already.
            mine = valid ? mine : N;
            for (int kk = 0; kk < kn; kk += 16) {      // four loads (16 neighbour rows) in flight per lane
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idu = __shfl(mine, hbase + kk + u * 4 + slot, 64);
                    const unsigned off = (__umul24((unsigned)idu, (unsigned)ldx) + (unsigned)c4) * 4u;
                    typedef unsigned hd_u4 __attribute__((ext_vector_type(4)));
                    const hd_u4 w = __builtin_amdgcn_raw_buffer_load_b128(xbuf, (int)off, 0, 0);
                    v[u] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                }
                const hd_f2 r2 = {rden, rden};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // neighbours of a point belong to the point's own cloud (the searches are per batch element): same den
                    const hd_f2 v01 = {v[u].x, v[u].y}, v23 = {v[u].z, v[u].w};
                    sum01 = sum01 + v01 * r2;
                    sum23 = sum23 + v23 * r2;
                }
            }
        } else {
            if (!valid) mine = -1;
            for (int kk = 0; kk < kn; kk += 16) {
                int id[4];
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kq = kk + u * 4 + slot;
                    id[u] = __shfl(mine, hbase + min(kq, 31), 64);
                    if (kq >= kn) id[u] = -1;
                    const float4 t = *(const float4*)(x + ((size_t)max(id[u], 0) * ldx + c4));
                    v[u] = id[u] < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : t;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float r = id[u] < 0 ? 0.f : rden;
                    const hd_f2 r2 = {r, r}, v01 = {v[u].x, v[u].y}, v23 = {v[u].z, v[u].w};
                    sum01 = sum01 + v01 * r2;
                    sum23 = sum23 + v23 * r2;
                }
            }
        }
    }
    float4 sum = make_float4(sum01.x, sum01.y, sum23.x, sum23.y);
    // combine the four row slots (lanes l, l^8, l^16, l^24 hold the same channels)
    sum.x += __shfl_xor(sum.x, 8, 64); sum.y += __shfl_xor(sum.y, 8, 64); sum.z += __shfl_xor(sum.z, 8, 64); sum.w += __shfl_xor(sum.w, 8, 64);
    sum.x += __shfl_xor(sum.x, 16, 64); sum.y += __shfl_xor(sum.y, 16, 64); sum.z += __shfl_xor(sum.z, 16, 64); sum.w += __shfl_xor(sum.w, 16, 64);
    const float fc = (float)max(cnt, 1);        // (cnt is already the half-wave's total: counted by ballot)
    float ymax = fmaxf(fmaxf(yv.x, yv.y), fmaxf(yv.z, yv.w));
    ymax = fmaxf(ymax, __shfl_xor(ymax, 1, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, 2, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, 4, 64));
    // The four row slots of a point hold the SAME four channel sums now: slot s evaluates channel 4*cl + s only (round 6).  The softplus
    // / divide chain is ~130 vector instructions per channel; evaluated for all four channels by all four slots it was 3/4 of the kernel's
    // vector work, and the kernel is issue-bound (104 M vector instructions per launch at F = 4 = 170 of its 182 us).  Same operations per
    // element, the maximum over a point's 32 channels is order-free: bit-identical scores.
    const float yj = slot == 0 ? yv.x : slot == 1 ? yv.y : slot == 2 ? yv.z : yv.w;
    const float sj = slot == 0 ? sum.x : slot == 1 ? sum.y : slot == 2 ? sum.z : sum.w;
    const float d = yj - sj / fc;
    float sp;   // softplus as TF computes it: log1p(exp(d)) with the large/small-argument shortcuts
    if (d > 15.f) sp = d;
    else if (d < -15.f) sp = expf(d);
    else sp = log1pf(expf(d));
    float best = sp * (yj / (1e-6f + ymax));
    best = fmaxf(best, __shfl_xor(best, 1, 64)); best = fmaxf(best, __shfl_xor(best, 2, 64)); best = fmaxf(best, __shfl_xor(best, 4, 64));
    best = fmaxf(best, __shfl_xor(best, 8, 64)); best = fmaxf(best, __shfl_xor(best, 16, 64));
    if (active && slot == 0) {
        const float inv = rsqrtf(fmaxf(sq, 1e-10f));
        *(float4*)&desc[(size_t)n * ldd + c4] = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
        if (l == 0) score[n] = best;
    }
}

extern "C" int d3f_detect_head(const float* x, int N, int ldx, int C, const int* idx, int ld_idx, int K,
                               const int* lens_dev, const int* include_zero_dev, int stack_group, int B, float* desc, int ldd,
                               float* score, int* scratch_dev, const int* row_order, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || C < 1 || C > 128 || ldx < C || ldd < C || K < 0 || ld_idx < K || B < 1 || B > D3F_MAX_BATCH || stack_group < 0)
        return D3F_ERR_ARG;
    if (N == 0) return D3F_OK;
    if (!x || !idx || !lens_dev || !desc || !score || !scratch_dev) return D3F_ERR_ARG;
    unsigned* mx = (unsigned*)scratch_dev;  // [B]
    int* offs = scratch_dev + B;            // [B+1]
    head_max_init_kernel<<<1, 256, 0, stream>>>(lens_dev, include_zero_dev, stack_group, B, mx, offs);
    // workgroups per cloud: every one ends in an atomicMax on the cloud's word -- same-address atomics serialise in L2, so as few
    // as still fill the chip (B x chunks ~ 512) rather than as many as the rows allow
    int chunks = d3f_cdiv((long long)N * C, 256 * 16);
    const int fill = d3f_cdiv(512, B);
    if (chunks > fill) chunks = fill;
    if (chunks < 1) chunks = 1;
    if (ldx == C && C % 4 == 0 && ((uintptr_t)x & 15) == 0) head_max_kernel<true><<<dim3(chunks, B), 256, 0, stream>>>(x, N, ldx, C, offs, B, mx);
    else head_max_kernel<false><<<dim3(chunks, B), 256, 0, stream>>>(x, N, ldx, C, offs, B, mx);
    const int blocks = d3f_cdiv((long long)N * 32, 256);
    const bool vec32 = C == 32 && ldx % 4 == 0 && ldd % 4 == 0 && (((uintptr_t)x | (uintptr_t)desc) & 15) == 0;
    if (vec32) {
        // one flag byte per row, behind the 2 B + 2 scratch words
        unsigned char* nz = (unsigned char*)(scratch_dev + 2 * B + 2);
        head32_rowflag_kernel<<<d3f_cdiv(N, 256), 256, 0, stream>>>(x, N, ldx, offs, B, mx, nz);
        if (d3f_fits_u24(N + 1, ldx) && ((long long)N + 1) * ldx < (1ll << 30))      // (row N, one past the end, is the shadow's address)
            head32_kernel<true><<<blocks, 256, 0, stream>>>(x, N, ldx, idx, ld_idx, K, offs, B, mx, nz, desc, ldd, score, row_order);
        else
            head32_kernel<false><<<blocks, 256, 0, stream>>>(x, N, ldx, idx, ld_idx, K, offs, B, mx, nz, desc, ldd, score, row_order);
    }
    else if (C <= 32) head_kernel<1><<<blocks, 256, 0, stream>>>(x, N, ldx, C, idx, ld_idx, K, offs, B, mx, desc, ldd, score, row_order);
    else if (C <= 64) head_kernel<2><<<blocks, 256, 0, stream>>>(x, N, ldx, C, idx, ld_idx, K, offs, B, mx, desc, ldd, score, row_order);
    else head_kernel<4><<<blocks, 256, 0, stream>>>(x, N, ldx, C, idx, ld_idx, K, offs, B, mx, desc, ldd, score, row_order);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone epilogue: out = act(x * col_scale + col_shift + residual)  (network_blocks.py:149-160, :185-186)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) affine_act_kernel(const float* __restrict__ x, int ldx, int M, int N,
                                                         const float* __restrict__ cs, const float* __restrict__ ch,
                                                         const float* __restrict__ res, int ldr, int leaky, float alpha,
                                                         float* __restrict__ out, int ldo, const int* __restrict__ M_dev) {
    M = d3f_dyn(M, M_dev);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)M * N) return;
    const int m = (int)(t / N), n = (int)(t % N);
    float v = x[(size_t)m * ldx + n];
    if (cs) v *= cs[n];
    if (ch) v += ch[n];
    if (res) v += res[(size_t)m * ldr + n];
    if (leaky) v = v > 0.f ? v : v * alpha;
    out[(size_t)m * ldo + n] = v;
}

extern "C" int d3f_affine_act(const float* x, int ldx, int M, int N, const float* col_scale, const float* col_shift,
                              const float* residual, int ldr, int leaky, float alpha, float* out, int ldo,
                              const int* M_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || N < 0 || ldx < N || ldo < N || (residual && ldr < N)) return D3F_ERR_ARG;
    if (M == 0 || N == 0) return D3F_OK;
    if (!x || !out) return D3F_ERR_ARG;
    affine_act_kernel<<<d3f_cdiv((long long)M * N, 256), 256, 0, stream>>>(x, ldx, M, N, col_scale, col_shift, residual, ldr,
                                                                          leaky, alpha, out, ldo, M_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// (xyz, desc, score) -> one record f32[3 + C + 1] per point: the 144-byte payload (C = 32) that the testers write per
// fragment (utils/tester.py:215-229 keeps points / features / scores together) and that the multi-GPU runner gathers once
// at the end.  One thread per output element; rows are contiguous, so a fragment's records are ONE contiguous block.
// ------------------------------------------------------------------------------------------------
// One lane per 16 output bytes: a block of rows is one contiguous run of the output, so the stores of a wavefront are full lines
// (one thread per 4-byte element with two integer divisions was 60-80 us per call for the 2 M rows of an eight-fragment stack).
// Destinations: rows are written to `out` (row n at n * ldo) unless the row belongs to one of the first `keep` clouds of a fragment
// whose entry in dst_ptrs is non-zero -- then to that address, the fragment's kept rows packed from its row 0: a replayed sequence
// puts a fragment's records where its consumer wants them (the rank's shard), no copy afterwards.  A fragment = `group` consecutive
// clouds of the stack (lens_dev: B entries).
__global__ void __launch_bounds__(256) pack_rows_kernel(const float* __restrict__ xyz, const float* __restrict__ desc, int ldd,
                                                        int C, const float* __restrict__ score, int N,
                                                        const int* __restrict__ N_dev, float* __restrict__ out, int ldo,
                                                        const int* __restrict__ lens_dev, int B, int group, int keep,
                                                        const long long* __restrict__ dst_ptrs, const int* __restrict__ row_map) {
    __shared__ int sStart[D3F_MAX_BATCH + 1];
    N = d3f_dyn(N, N_dev);
    if (dst_ptrs) {
        if (threadIdx.x == 0) {
            int s = 0;
            for (int b = 0; b < B; ++b) { sStart[b] = s; s += lens_dev[b]; }
            sStart[B] = s;
        }
        __syncthreads();
    }
    const int W = C + 4, Q = W >> 2;                    // floats / 16-byte pieces per record (W % 4 == 0: checked by the launcher)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * Q) return;
    const int n = (int)(t / Q), q = (int)(t - (long long)n * Q);
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = 4 * q + u;
        v[u] = c < 3 ? xyz[3 * (size_t)n + c] : (c < 3 + C ? desc[(size_t)n * ldd + (c - 3)] : score[n]);
    }
    // row_map: the inputs are in an internal row order, record n belongs at row row_map[n] (a cloud's rows stay inside its range)
    const int no = row_map ? row_map[n] : n;
    float* dst = out + (size_t)no * ldo + 4 * q;
    if (dst_ptrs) {
        int b = 0;
        while (b + 1 < B && no >= sStart[b + 1]) ++b;
        const int f = b / group;
        const long long base = dst_ptrs[f];
        if (base != 0 && b - f * group < keep) dst = (float*)base + (size_t)(no - sStart[f * group]) * ldo + 4 * q;
    }
    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
}

// any record width (one thread per element): descriptor widths that are not a multiple of four floats
__global__ void __launch_bounds__(256) pack_rows_scalar_kernel(const float* __restrict__ xyz, const float* __restrict__ desc, int ldd,
                                                               int C, const float* __restrict__ score, int N,
                                                               const int* __restrict__ N_dev, float* __restrict__ out, int ldo) {
    N = d3f_dyn(N, N_dev);
    const int W = C + 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * W) return;
    const int n = (int)(t / W), c = (int)(t % W);
    out[(size_t)n * ldo + c] = c < 3 ? xyz[3 * (size_t)n + c] : (c < 3 + C ? desc[(size_t)n * ldd + (c - 3)] : score[n]);
}

static int pack_launch(const float* xyz, const float* desc, int ldd, int C, const float* score, int N, float* out, int ldo,
                       const int* N_dev, const int* lens_dev, int B, int group, int keep, const long long* dst_ptrs, const int* row_map,
                       hipStream_t stream) {
    if (N < 0 || C < 1 || ldd < C || ldo < C + 4) return D3F_ERR_ARG;
    if (N == 0) return D3F_OK;
    if (!xyz || !desc || !score || !out) return D3F_ERR_ARG;
    if (((C + 4) & 3) || (ldo & 3) || ((uintptr_t)out & 15)) {
        if (dst_ptrs || row_map) return D3F_ERR_ARG;           // (per-fragment destinations / row maps: 16-byte pieces only)
        pack_rows_scalar_kernel<<<d3f_cdiv((long long)N * (C + 4), 256), 256, 0, stream>>>(xyz, desc, ldd, C, score, N, N_dev, out, ldo);
        D3F_LAUNCH_CHECK();
        return D3F_OK;
    }
    pack_rows_kernel<<<d3f_cdiv((long long)N * ((C + 4) / 4), 256), 256, 0, stream>>>(xyz, desc, ldd, C, score, N, N_dev, out, ldo, lens_dev,
                                                                                   B, group, keep, dst_ptrs, row_map);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" int d3f_pack_descriptors(const float* xyz, const float* desc, int ldd, int C, const float* score, int N, float* out,
                                    int ldo, const int* N_dev, void* stream_) {
    return pack_launch(xyz, desc, ldd, C, score, N, out, ldo, N_dev, nullptr, 0, 1, 0, nullptr, nullptr, (hipStream_t)stream_);
}

extern "C" int d3f_pack_descriptors_to(const float* xyz, const float* desc, int ldd, int C, const float* score, int N, float* out,
                                       int ldo, const int* N_dev, const int* lens_dev, int B, int group, int keep,
                                       const long long* dst_ptrs_dev, const int* row_map_dev, void* stream_) {
    if (dst_ptrs_dev && (!lens_dev || B < 1 || B > D3F_MAX_BATCH || group < 1 || keep < 0 || keep > group)) return D3F_ERR_ARG;
    if (!dst_ptrs_dev && !row_map_dev) return D3F_ERR_ARG;
    return pack_launch(xyz, desc, ldd, C, score, N, out, ldo, N_dev, lens_dev, B, group, keep, dst_ptrs_dev, row_map_dev,
                       (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------
// Stage-0 ingestion: xyz out of raw file records on the GPU.  A binary PLY vertex element (demo_registration.py:23,
// datasets/ThreeDMatch.py:348: float or double coordinates among other properties, either byte order) or a KITTI velodyne
// sweep (datasets/KITTI.py:131: 16-byte records) is copied to the device AS BYTES; one thread per point picks the three
// coordinates out of its record and writes the float32 [n,3] array the grid subsampler reads -- the host never touches the
// payload.  double -> float is the round-to-nearest cast numpy's astype(float32) performs in the reference's loaders.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rec_coord(const unsigned char* __restrict__ p, int is_f64, int swap) {
    if (is_f64) {
        unsigned long long u = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) u |= (unsigned long long)p[swap ? 7 - b : b] << (8 * b);
        return (float)__longlong_as_double((long long)u);
    }
    unsigned u = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) u |= (unsigned)p[swap ? 3 - b : b] << (8 * b);
    return __uint_as_float(u);
}

__global__ void __launch_bounds__(256) decode_records_kernel(const unsigned char* __restrict__ raw, int n, int stride, int ox,
                                                             int oy, int oz, int is_f64, int swap, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* r = raw + (size_t)i * stride;
    out[3 * (size_t)i] = rec_coord(r + ox, is_f64, swap);
    out[3 * (size_t)i + 1] = rec_coord(r + oy, is_f64, swap);
    out[3 * (size_t)i + 2] = rec_coord(r + oz, is_f64, swap);
}

extern "C" int d3f_decode_xyz_records(const void* raw, int n, int stride, int off_x, int off_y, int off_z, int is_f64,
                                      int big_endian, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int w = is_f64 ? 8 : 4;
    if (n < 0 || stride < 3 * w || off_x < 0 || off_y < 0 || off_z < 0 || off_x + w > stride || off_y + w > stride ||
        off_z + w > stride)
        return D3F_ERR_ARG;
    if (n == 0) return D3F_OK;
    if (!raw || !out) return D3F_ERR_ARG;
    decode_records_kernel<<<d3f_cdiv(n, 256), 256, 0, stream>>>((const unsigned char*)raw, n, stride, off_x, off_y, off_z,
                                                                is_f64 ? 1 : 0, big_endian ? 1 : 0, out);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

extern "C" int d3f_version(void) { return 200; }
