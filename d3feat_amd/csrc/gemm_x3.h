// fp32 contraction on the bf16 matrix cores by EXACT operand splitting (included by gemm_f32.hip).
//
// An fp32 value has 24 significant bits; three bfloat16 values hold 8 each.  a = a1 + a2 + a3 with
//     a1 = bf16(a),  a2 = bf16(a - a1),  a3 = bf16(a - a1 - a2)          (both differences are exact in fp32)
// is an exact identity (round-to-nearest pieces, |a - a1 - a2| < 2^-17 |a| has at most 7 significant bits left), the same for w.
// A product of two bf16 values is exact in fp32 (8 x 8 = 16 bits), so
//     a w = a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)  +  [ a2 w3 + a3 w2 + a3 w3 : |.| < 2^-23 |a w| ]
// -- six exact products per fp32 product, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16 k per instruction, 32 cycles)
// instead of one rounded product on v_mfma_f32_32x32x2_f32 (2 k per instruction, 64 cycles): 2.67x the matrix-pipe rate and a
// sixth of the k-steps a wavefront has to issue instructions around.  What is dropped is below the rounding of the fp32
// accumulation itself; tests/test_gpu_gemm_x3.py measures the result against float64 next to the fp32 MFMA kernel's.
//
// Layout.  Workgroup = 4 wavefronts stacked along M: 128 rows x 32 TN columns; wave w owns rows 32 w .. 32 w + 31 and ALL the
// workgroup's columns (TN accumulator tiles), so no A element is loaded or split twice inside a workgroup.
//   A  never touches LDS: the MFMA A operand wants, per lane (row = lane & 31, half = lane >> 5), eight consecutive k of one
//      row -- 32 contiguous bytes of the fp32 row.  Each lane loads them straight from global memory (two float4 per 16-deep
//      step, one k-tile ahead), splits them in registers (v_cvt_pk_bf16_f32 + shift / mask + subtract) into the three operand
//      planes.  Gathered / concatenated operands ([ x'[idx[m]] | skip[m] ]) are a per-lane row pointer.
//   W  pre-split ONCE (d3f_gemm_pack_x3): bf16 [column group of 32][k-tile of 32][plane 3][32 rows][40] -- the LDS image itself,
//      rows padded to 80 bytes (the 16 lanes of a ds_read_b128 group on 16 distinct 4-bank slots), so staging a k-tile is a
//      linear 7680-byte copy per column group (global -> registers -> ds_write_b128), double buffered, one barrier per k-tile.
// Per k-tile and wave: 4 global loads, ~90 vector ALU instructions of splitting, 6 TN ds_read_b128, 12 TN MFMAs (384 TN cycles).
#pragma once

#define GX_BK 32
#define GX_LS 40                      // bf16 per LDS row
#define GX_CHUNK (3 * 32 * GX_LS)     // bf16 per (column group, k-tile) chunk = 3840 (7680 bytes, 480 uint4)

__device__ __forceinline__ unsigned gx_cvt_pk(float lo, float hi) {     // two fp32 -> two bf16 (RNE), lo in bits 0..15
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// Operand requests are issued through inline asm and waited for by hand (the results are tied through the s_waitcnt statement, so no
// consumer can be scheduled above it): left to the compiler, the loads of the NEXT k-tile -- written before this tile's MFMAs --
// are sunk to their first use at the top of the next iteration and the whole memory latency is exposed once per k-tile.
typedef float gx_f4 __attribute__((ext_vector_type(4)));
typedef unsigned gx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gx_ld16(gx_f4& r, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
__device__ __forceinline__ void gx_ld16o(gx_f4& r, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r) : "v"(p) : "memory");
}
__device__ __forceinline__ void gx_ld16(gx_u4& r, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }

// eight consecutive-k floats -> the three operand planes (eight bf16 = one uint4 each)
__device__ __forceinline__ void gx_split8(const gx_f4& x0, const gx_f4& x1, uint4 (&pl)[3]) {
    float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned p[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        p[0][i] = gx_cvt_pk(a, b);
        a -= __uint_as_float(p[0][i] << 16);
        b -= __uint_as_float(p[0][i] & 0xffff0000u);
        p[1][i] = gx_cvt_pk(a, b);
        a -= __uint_as_float(p[1][i] << 16);
        b -= __uint_as_float(p[1][i] & 0xffff0000u);
        p[2][i] = gx_cvt_pk(a, b);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) pl[s] = make_uint4(p[s][0], p[s][1], p[s][2], p[s][3]);
}

// W f32[K,N] (ldb) -> Wx (layout above); K rows beyond K and columns beyond N are zero
__global__ void __launch_bounds__(256) gemm_pack_x3_kernel(const float* __restrict__ B, int ldb, int K, int N, int nkt, long long total,
                                                           unsigned short* __restrict__ Wx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int kk = (int)(t % GX_LS);
    long long q = t / GX_LS;
    const int r = (int)(q % 32); q /= 32;
    const int p = (int)(q % 3); q /= 3;
    const int kt = (int)(q % nkt);
    const int ng = (int)(q / nkt);
    const int k = kt * GX_BK + kk, n = ng * 32 + r;
    unsigned out = 0u;
    if (kk < GX_BK && k < K && n < N) {
        float w = B[(size_t)k * ldb + n];
        unsigned h = gb_rne(w);
        if (p > 0) { w -= __uint_as_float(h << 16); h = gb_rne(w); }
        if (p > 1) { w -= __uint_as_float(h << 16); h = gb_rne(w); }
        out = h;
    }
    Wx[t] = (unsigned short)out;
}

template <int TN>
#ifndef GX_EXP_WAVES
#define GX_EXP_WAVES 2, 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GX_EXP_WAVES)))
gemm_x3_kernel(const float* __restrict__ A, int lda, const unsigned short* __restrict__ Wx, int nkt, int NG, float* __restrict__ C,
               int ldc, int M, int N, int tiles_per_split, float* __restrict__ slab, GemmEpi E, const int* __restrict__ M_dev,
               GemmGather G) {
    constexpr int BM = 128, BN = 32 * TN;
    constexpr int NB = (480 * TN + 255) / 256;      // uint4 of W per thread and k-tile
    const int Mcap = M;
    M = d3f_dyn(M, M_dev);
    if ((int)(blockIdx.z * BM) >= M) return;        // capacity-sized grid (row tile = slowest dispatch dimension)
    // live workgroups = the first contiguous run of the dispatch order; every XCD takes one contiguous run of (row tile, K slice,
    // column tile) triples, so the column tiles of a row block -- which read the same A rows -- meet in one L2
    const unsigned gx_ = gridDim.x, gxy_ = gridDim.x * gridDim.y;
    const unsigned T_ = d3f_xcd_tile(blockIdx.x + gx_ * blockIdx.y + gxy_ * blockIdx.z, gxy_ * (unsigned)((M + BM - 1) / BM));
    const unsigned bz = T_ / gxy_, by = (T_ % gxy_) / gx_, bx = T_ % gx_;
    // (+ 512: the last uint4 round of the staging copy is not full; its surplus threads store into this tail instead of branching)
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][TN * GX_CHUNK + 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bz * BM, n0 = bx * BN;
    const int t_begin = by * tiles_per_split;
    const int t_end = min(nkt, t_begin + tiles_per_split);

    // ---- the lane's A row: its own, or the gathered one; second operand of a concatenation.  Branch-free loads: a row that does not
    // exist (beyond M, shadow / out-of-range index) is the zero line read at offset 0 (offset mask 0).
    const int gm_a = m0 + 32 * wave + (lane & 31);
    const float* arow = gd_zero_line;
    const float* a2row = gd_zero_line;
    unsigned amask = 0u, a2mask = 0u;
    if (gm_a < M) {
        int sr = gm_a;
        if (G.gidx) {
            const int n1 = d3f_dyn(G.N1, G.N1_dev);
            sr = G.gidx[(size_t)gm_a * G.ld_gidx];
            if (sr < 0 || sr >= n1) sr = -1;
        }
        if (sr >= 0) { arow = A + (size_t)sr * lda; amask = 0xffffffffu; }
        if (G.A2) { a2row = G.A2 + (size_t)gm_a * G.lda2; a2mask = 0xffffffffu; }
    }
    const int kofs = (lane >> 5) << 3;
    const int K1 = G.K1;                                   // a multiple of 32 when there is a second operand: a k-tile has ONE source

    // ---- the thread's share of a W k-tile: uint4 number e = tid + 256 i of the TN chunks of 480 (surplus: a valid chunk, LDS tail).
    // (named scalars, not arrays: an array of uint4 that lives across the loop is demoted to LDS by the compiler)
    auto wsrc = [&](int i) {
        const int e = tid + 256 * i;
        const int c = e / 480, o = e - 480 * c;
        const int ng = min((int)bx * TN + c, NG - 1);
        return (const uint4*)(Wx + ((size_t)ng * nkt + t_begin) * GX_CHUNK) + o;
    };
    auto wdst = [&](int i) {
        const int e = tid + 256 * i;
        const int c = e / 480, o = e - 480 * c;
        return c * GX_CHUNK + o * 8;
    };
    const uint4 *bsrc0 = wsrc(0), *bsrc1 = wsrc(1), *bsrc2 = wsrc(NB > 2 ? 2 : 0), *bsrc3 = wsrc(NB > 3 ? 3 : 0);
    const int bdst0 = wdst(0), bdst1 = wdst(1), bdst2 = wdst(2), bdst3 = wdst(3);
    gx_u4 rb0, rb1, rb2, rb3;
    gx_f4 raw00, raw01, raw10, raw11;          // [16-deep step][float4 of the lane's eight k]
    uint4 ap[2][3];
    // W first, A second: the staging stores need only the older half of the queue (vmcnt(4))
    auto request = [&](int t) {
        const size_t wo = (size_t)(t - t_begin) * (GX_CHUNK / 8);
        gx_ld16(rb0, bsrc0 + wo);
        gx_ld16(rb1, bsrc1 + wo);
        if (NB > 2) gx_ld16(rb2, bsrc2 + wo);
        if (NB > 3) gx_ld16(rb3, bsrc3 + wo);
        const bool first = t * GX_BK < K1;                 // (wave-uniform)
        const float* base = first ? arow : a2row;
        const unsigned msk = first ? amask : a2mask;
        const int kk = (first ? t * GX_BK : t * GX_BK - K1) + kofs;
        const float* p0 = base + ((unsigned)kk & msk);
        const float* p1 = base + ((unsigned)(kk + 16) & msk);
#ifdef GX_EXP_NOLOADA
        gx_ld16(raw00, gd_zero_line);
        gx_ld16(raw01, gd_zero_line);
        gx_ld16(raw10, gd_zero_line);
        gx_ld16(raw11, gd_zero_line);
        asm volatile("" : : "v"(p0), "v"(p1));
#else
        gx_ld16(raw00, p0);
        gx_ld16o(raw01, p0);
        gx_ld16(raw10, p1);
        gx_ld16o(raw11, p1);
#endif
    };
    auto store_b = [&](int buf) {
        if (NB > 3) asm volatile("s_waitcnt vmcnt(4)" : "+v"(rb0), "+v"(rb1), "+v"(rb2), "+v"(rb3) : : "memory");
        else if (NB > 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(rb0), "+v"(rb1), "+v"(rb2) : : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(rb0), "+v"(rb1) : : "memory");
        *(gx_u4*)&Bs[buf][bdst0] = rb0;
        *(gx_u4*)&Bs[buf][bdst1] = rb1;
        if (NB > 2) *(gx_u4*)&Bs[buf][bdst2] = rb2;
        if (NB > 3) *(gx_u4*)&Bs[buf][bdst3] = rb3;
    };
    auto split = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw00), "+v"(raw01), "+v"(raw10), "+v"(raw11) : : "memory");
#ifdef GX_EXP_NOSPLIT
        ap[0][0] = __builtin_bit_cast(uint4, raw00); ap[0][1] = __builtin_bit_cast(uint4, raw01); ap[0][2] = ap[0][0];
        ap[1][0] = __builtin_bit_cast(uint4, raw10); ap[1][1] = __builtin_bit_cast(uint4, raw11); ap[1][2] = ap[1][0];
#else
        gx_split8(raw00, raw01, ap[0]);
        gx_split8(raw10, raw11, ap[1]);
#endif
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // W fragments: lane (c = lane & 31, h = lane >> 5) holds k = 8 h .. 8 h + 7 of a 16-deep step, row c of the column group
    auto compute = [&](int buf) {
        const unsigned short* bp = &Bs[buf][(lane & 31) * GX_LS + kofs];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 b[TN][3];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = *(const uint4*)(bp + j * GX_CHUNK + p * (32 * GX_LS) + 16 * s);
            // smallest terms first; the TN accumulator chains alternate
#define GX_MFMA(PA_, PB_)                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                             \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, b[j][PB_]),                             \
                                                         __builtin_bit_cast(gb_bf16x8, ap[s][PA_]), acc[j], 0, 0, 0)
#ifndef GX_EXP_ONEMFMA
            GX_MFMA(2, 0);
            GX_MFMA(1, 1);
            GX_MFMA(0, 2);
            GX_MFMA(1, 0);
            GX_MFMA(0, 1);
#endif
            GX_MFMA(0, 0);
#undef GX_MFMA
        }
    };

    // One k-tile per iteration: the tile's operands were requested a whole compute phase ago (raw A floats and W in registers);
    // they are staged / split at the top, one barrier publishes W (and fences the buffer the stores of the NEXT iteration reuse),
    // the next tile's requests are issued and fly during this tile's 12 TN MFMAs.  (The last iteration re-requests its own tile:
    // no branch in the loop; the final wait keeps the request registers alive until those loads have landed.)
    if (t_begin < t_end) {
        request(t_begin);
        int buf = 0;
        for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
            store_b(buf);
            split();
            __syncthreads();
            request(min(t + 1, t_end - 1));
            compute(buf);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb0), "+v"(rb1), "+v"(raw00), "+v"(raw01), "+v"(raw10), "+v"(raw11) : : "memory");
        if (NB > 2) asm volatile("" : "+v"(rb2) : : "memory");
        if (NB > 3) asm volatile("" : "+v"(rb3) : : "memory");
    }

    // The product is computed transposed (W fragment as the MFMA's first operand): D[i][jm] with i = column of the group, jm = the
    // lane's own A row.  C/D layout of the 32x32 MFMA: jm = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- four consecutive
    // output columns per register quad: 16-byte stores, and the row's scale / residual belong to the lane that loaded the row.
    const bool mok = gm_a < M;
    float rs = 1.f;
    if (!slab && E.row_scale && mok) rs = E.row_scale[gm_a];
    float* dst = slab ? slab + ((size_t)by * Mcap + (mok ? gm_a : 0)) * N : C + (size_t)(mok ? gm_a : 0) * ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gn = n0 + 32 * j + 8 * q + 4 * (lane >> 5);
            if (!mok || gn >= N) continue;           // (N is a multiple of 4: a quad is inside or outside)
            float v[4] = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
            if (!slab) {
                const float4 c4 = E.col_scale ? *(const float4*)&E.col_scale[gn] : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 h4 = E.col_shift ? *(const float4*)&E.col_shift[gn] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 r4 = E.residual ? *(const float4*)&E.residual[(size_t)gm_a * E.ldr + gn] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float c[4] = {c4.x, c4.y, c4.z, c4.w}, h[4] = {h4.x, h4.y, h4.z, h4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (v[e] * rs) * c[e] + h[e];
                    t += rr[e];
                    v[e] = (E.leaky && !(t > 0.f)) ? t * E.alpha : t;
                }
            }
            *(float4*)&dst[gn] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}
