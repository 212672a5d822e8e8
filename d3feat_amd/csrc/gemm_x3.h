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
// Edges of the format (tests: the e^+-20 exactness cases, test_the_small_edge_of_the_format): an operand that is +-Inf, or finite
// within 2^-8 of FLT_MAX (its first plane rounds to Inf), yields NaN (Inf - Inf in the second plane) where the fp32 pipe yields
// +-Inf -- non-finite either way; planes that are bfloat16 SUBNORMALS (the third plane of operands below ~2^-110, the second below
// ~2^-118) are flushed by the matrix pipe: measured max abs error 4.6e-41 = 2^-134 on operands of 2^-126 .. 2^-90, nothing flushed
// to zero, operands of 2^-100 and above bit for bit.
// Toolchain contract: the A loads and the LDS-DMA are inline asm whose outputs land after the statement; nothing may
// copy or spill those registers between request and wait, and the hand-counted vmcnt assumes no compiler-issued VMEM inside the
// k-loop.  Checked after every change of this file: -Rpass-analysis=kernel-resource-usage reports 0 spilled VGPRs for every
// gemm_x3 / gemm_x3r instance (a spill did appear in round 5 -- of a lane constant, not of a request register -- and cost a
// vmcnt(0) per tile: see gemm_x3r_kernel), and the exact-plane tests are part of the GPU suite.
//
// Layout.  Workgroup = 4 wavefronts stacked along M: 128 rows x 32 TN columns; wave w owns rows 32 w .. 32 w + 31 and ALL the
// workgroup's columns (TN accumulator tiles), so no A element is loaded or split twice inside a workgroup.
//   A  never touches LDS: the MFMA operand wants, per lane (row = lane & 31, half = lane >> 5), eight consecutive k of one row --
//      32 contiguous bytes of the fp32 row.  Each lane requests them straight from global memory (four float4 per k-tile, TWO
//      k-tiles ahead, two register sets) and splits them in registers into the three operand planes (v_cvt_pk_bf16_f32, shift /
//      mask, packed subtract: 72 vector instructions per k-tile).  Gathered / concatenated operands ([ x'[idx[m]] | skip[m] ])
//      are a per-lane row pointer; rows that do not exist read a zero line.
//   W  pre-split ONCE (d3f_gemm_pack_x3): bf16 [column group of 32][k-tile of 32][plane 3][32 rows][40] -- the LDS image itself,
//      rows padded to 80 bytes (the 16 lanes of a ds_read_b128 group on 16 distinct 4-bank slots), so staging a k-tile is a
//      linear copy of 7680 bytes per column group: LDS-DMA (global_load_lds_dwordx4) into a ring of three slots, two k-tiles
//      ahead, no staging registers and no ds_write pass; one barrier per k-tile.
// Per k-tile and wave: 4 global loads + NB DMA pieces, 72 splitting instructions, 6 TN ds_read_b128, 12 TN MFMAs (384 TN cycles).
//
// Measured (profiles/r04_experiments.txt x1-x8): error against float64 BELOW the fp32 MFMA kernel's on the same operands (3e-7 vs
// 4.4e-7 of max |C|); 1.15x the LDS-DMA fp32 kernel over the network's 26 launches, 1.3-1.4x on its K-long ones.  On gfx950 vector
// ALU work does NOT hide behind MFMAs of other wavefronts of the SIMD (tools/ubench/mfma_valu_overlap.hip: 24 MFMA + 120 VALU per
// wave cost 0.48 us where the MFMAs alone cost 0.33, whatever the number of waves), which bounds this form at ~0.7 of the bf16
// matrix rate it issues; the network's launches are short of that mostly for their size (<= 14 GFLOP, 2-24 k-tiles per workgroup).
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
__device__ __forceinline__ void gx_ld16(gx_f4& r, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
template <int OFF> __device__ __forceinline__ void gx_ld16o(gx_f4& r, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}

// eight consecutive-k floats -> the three operand planes (eight bf16 = one uint4 each).  Per pair of floats: v_cvt_pk_bf16_f32, the
// two pieces widened again (shift / mask), two subtracts, twice, and a last v_cvt_pk: 11 vector instructions.  The subtracts go
// through inline asm: left alone, the SLP vectoriser packs them into v_pk_add_f32, which costs ~13 extra cycles each beside MFMAs
// (MI355X_MICROARCH.md "price of one filler beside MFMAs"; tools/ubench/mfma_valu_overlap.hip).
__device__ __forceinline__ float gx_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void gx_split8(const gx_f4& x0, const gx_f4& x1, uint4 (&pl)[3]) {
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned p[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        p[0][i] = gx_cvt_pk(a, b);
        a = gx_sub(a, __uint_as_float(p[0][i] << 16));
        b = gx_sub(b, __uint_as_float(p[0][i] & 0xffff0000u));
        p[1][i] = gx_cvt_pk(a, b);
        a = gx_sub(a, __uint_as_float(p[1][i] << 16));
        b = gx_sub(b, __uint_as_float(p[1][i] & 0xffff0000u));
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

// One LDS-DMA round of a W k-tile: NB pieces per wave (64 lanes x 16 bytes each, linear from the wave's base M0; consecutive pieces
// of a workgroup are STRIDE = 1 KiB x its wavefronts apart), the sources addressed as scalar base + per-lane 32-bit byte offset.
// M0 is saved / restored around the round; the asm is absent from the compiler's vmcnt bookkeeping (see above).
template <int NB, int STRIDE>
__device__ __forceinline__ void gx_dma_round(const void* sbase, unsigned dst, const unsigned (&vo)[4]) {
    unsigned keep;
    if constexpr (NB == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "s"(dst), "v"(vo[0]), "v"(vo[1]), "n"(STRIDE) : "memory", "scc");
    else if constexpr (NB == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "s"(dst), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "n"(STRIDE) : "memory", "scc");
    else static_assert(NB < 0, "add the round");
}

// TN: 32-column groups per workgroup; WAVES: its wavefronts (32 rows each).  Shipped: 128 x 32, 128 x 64 (4 waves) and 256 x 128 (8)
template <int TN, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_x3_kernel(const float* __restrict__ A, int lda, const unsigned short* __restrict__ Wx, int nkt, int NG, float* __restrict__ C,
               int ldc, int M, int N, int tiles_per_split, float* __restrict__ slab, GemmEpi E, const int* __restrict__ M_dev,
               GemmGather G) {
    constexpr int BM = 32 * WAVES, BN = 32 * TN, NTH = 64 * WAVES;
    constexpr int NB = (480 * TN + NTH - 1) / NTH;  // DMA pieces (uint4 per thread) of a W k-tile: 2 or 4
    constexpr int SB = NB * NTH * 16;               // bytes per ring slot: the TN chunks + the tail the surplus lanes of the last piece hit
    constexpr int WAITN = 4 + NB;                   // requests of ONE iteration: what may still be in flight when a tile is consumed
    static_assert((TN == 1 || TN == 2 || TN == 4) && (WAVES == 4 || WAVES == 8), "tile shape");
    const int Mcap = M;
    M = d3f_dyn(M, M_dev);
    if ((int)(blockIdx.z * BM) >= M) return;        // capacity-sized grid (row tile = slowest dispatch dimension)
    // live workgroups = the first contiguous run of the dispatch order; every XCD takes one contiguous run of (row tile, K slice,
    // column tile) triples, so the column tiles of a row block -- which read the same A rows -- meet in one L2
    const unsigned gx_ = gridDim.x, gxy_ = gridDim.x * gridDim.y;
    const unsigned T_ = d3f_xcd_tile(blockIdx.x + gx_ * blockIdx.y + gxy_ * blockIdx.z, gxy_ * (unsigned)((M + BM - 1) / BM));
    const unsigned bz = T_ / gxy_, by = (T_ % gxy_) / gx_, bx = T_ % gx_;
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bz * BM, n0 = bx * BN;
    const int t_begin = by * tiles_per_split;
    const int t_end = min(nkt, t_begin + tiles_per_split);
    const int t_last = t_end - 1;

    // ---- the lane's A row: its own, or the gathered one; second operand of a concatenation.  Branch-free loads: a row that does not
    // exist (beyond M, shadow / out-of-range index) is the zero line, read at offset 0 whatever the k-tile (offset mask 0).
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
    // running pointer to the lane's eight floats of the next tile to REQUEST (requests go tile by tile, clamped at the last)
    const float* pa;
    unsigned astep;                                        // floats per k-tile for this lane: 32, or 0 on the zero line
    int treq = t_begin;
    auto a_rebase = [&](int t) {
        const bool first = t * GX_BK < K1;                 // (wave-uniform)
        const unsigned msk = first ? amask : a2mask;
        pa = (first ? arow : a2row) + ((unsigned)((first ? t * GX_BK : t * GX_BK - K1) + kofs) & msk);
        astep = (unsigned)GX_BK & msk;
    };
    a_rebase(t_begin);

    // ---- the thread's share of a W k-tile: uint4 number e = tid + 256 i of the TN chunks of 480; the LDS image of a slot is the
    // chunks one after the other, i.e. uint4 number e lands at byte 16 e: an LDS-DMA piece per wave and i.  (Surplus lanes of the
    // last piece: a valid chunk as source, the slot's tail as target.)  Source = scalar base of the k-tile + per-lane byte offset.
    unsigned wofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + NTH * (i < NB ? i : 0);
        const int c = e / 480, o = e - 480 * c;
        const int ng = min((int)bx * TN + c, NG - 1);
        wofs[i] = (unsigned)ng * (unsigned)nkt * (unsigned)(GX_CHUNK * 2) + (unsigned)o * 16u;
    }
    const unsigned lds_wave = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((unsigned)(unsigned long long)(gd_lptr)Bs + (unsigned)wave * 1024u));

    gx_f4 rA0, rA1, rA2, rA3, rB0, rB1, rB2, rB3, rC0, rC1, rC2, rC3;      // three k-tiles of the lane's A floats: two in flight
    uint4 ap0[2][3], ap1[2][3];                                           // operand planes of the current / the next k-tile
    auto request_w = [&](int t, int slot) {
        gx_dma_round<NB, WAVES * 1024>((const char*)Wx + (size_t)t * (GX_CHUNK * 2), lds_wave + (unsigned)slot * (unsigned)SB, wofs);
    };
    auto request_a = [&](gx_f4& r0, gx_f4& r1, gx_f4& r2, gx_f4& r3) {
        gx_ld16(r0, pa);
        gx_ld16o<16>(r1, pa);
        gx_ld16o<64>(r2, pa);
        gx_ld16o<80>(r3, pa);
        if (treq < t_last) {                               // (wave-uniform) advance to the next tile; past the end: the last tile again
            ++treq;
            if (treq * GX_BK == K1) a_rebase(treq);
            else pa += astep;
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // One k-tile: the 12 TN MFMAs of tile t on planes `cur`, and IN THEIR SHADOW the splitting of tile t + 1's floats (r0..r3) into
    // planes `nxt`.  On gfx950 vector instructions do not overlap with the MFMAs of OTHER wavefronts, only with the wave's own when
    // they are interleaved a few per MFMA (tools/ubench/mfma_valu_overlap.hip: 24 MFMA + 120 VALU cost 0.57 us one after the other,
    // 0.38 us interleaved 1 : 5, against 0.37 us for the MFMAs alone -- whatever the number of waves per SIMD).  The 88 splitting
    // instructions are therefore written as a flat list of micro-operations (seven dependent levels over the 16 floats) and dealt
    // out behind the MFMAs in program order, pinned by sched_barrier (the sched_group_barrier masks do not see inline asm).
    // W fragments: lane (c = lane & 31, h = lane >> 5) holds k = 8 h .. 8 h + 7 of a 16-deep step, row c of the column group.
    auto tile = [&](int slot, const uint4 (&cur)[2][3], uint4 (&nxt)[2][3], const gx_f4& r0, const gx_f4& r1, const gx_f4& r2,
                    const gx_f4& r3) {
        const unsigned short* bp = (const unsigned short*)(Bs + slot * SB) + (lane & 31) * GX_LS + kofs;
        constexpr int NM = 12 * TN, NOPS = 88;
        // (operand plane of A, of W) per product.  First 16-deep step: smallest terms first.  Second step: W planes in the order
        // 2, 1, 1, 0, 0, 0 -- its fragments are read into the registers of the first step's, plane by plane as the first step is
        // done with them (plane 2 after its third product, plane 1 after the fifth, plane 0 after the sixth): no second set of
        // fragment registers, which is what keeps three wavefronts per SIMD resident.
        constexpr int PA[2][6] = {{2, 1, 0, 1, 0, 0}, {0, 1, 0, 2, 1, 0}}, PB[2][6] = {{0, 1, 2, 0, 1, 0}, {2, 1, 1, 0, 0, 0}};
        uint4 b[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = *(const uint4*)(bp + j * GX_CHUNK + p * (32 * GX_LS));
        float x[16] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3], r3[0], r3[1], r3[2], r3[3]};
        float hl[16];
        unsigned P[3][8];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int s = m / (6 * TN), prod = (m % (6 * TN)) / TN, j = m % TN;        // the TN accumulator chains alternate
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, b[j][PB[s][prod]]),
                                                             __builtin_bit_cast(gb_bf16x8, cur[s][PA[s][prod]]), acc[j], 0, 0, 0);
            if (s == 0 && (prod == 2 || prod >= 4)) {          // this chain's last use of a first-step plane: refill it for the second
                const int p = prod == 2 ? 2 : prod == 4 ? 1 : 0;
                b[j][p] = *(const uint4*)(bp + j * GX_CHUNK + p * (32 * GX_LS) + 16);
            }
#pragma unroll
            for (int k = m * NOPS / NM; k < (m + 1) * NOPS / NM; ++k) {
                const int lvl = k < 8 ? 0 : k < 24 ? 1 : k < 40 ? 2 : k < 48 ? 3 : k < 64 ? 4 : k < 80 ? 5 : 6;
                if (lvl == 0) P[0][k] = gx_cvt_pk(x[2 * k], x[2 * k + 1]);
                else if (lvl == 3) P[1][k - 40] = gx_cvt_pk(x[2 * (k - 40)], x[2 * (k - 40) + 1]);
                else if (lvl == 6) P[2][k - 80] = gx_cvt_pk(x[2 * (k - 80)], x[2 * (k - 80) + 1]);
                else if (lvl == 1 || lvl == 4) {
                    const int e = k - (lvl == 1 ? 8 : 48);
                    const unsigned pk = P[lvl == 1 ? 0 : 1][e >> 1];
                    hl[e] = __uint_as_float((e & 1) ? (pk & 0xffff0000u) : (pk << 16));
                } else {
                    const int e = k - (lvl == 2 ? 24 : 64);
                    x[e] = gx_sub(x[e], hl[e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) nxt[s][p] = make_uint4(P[p][4 * s], P[p][4 * s + 1], P[p][4 * s + 2], P[p][4 * s + 3]);
    };
    // everything but the newest WAITN requests has landed (the queue retires in order): W of tile t and the A floats of tile t + 1
    auto landed = [&](gx_f4& r0, gx_f4& r1, gx_f4& r2, gx_f4& r3) {
        if constexpr (WAITN == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
        static_assert(WAITN == 6 || WAITN == 8, "wait count");
    };

    // Requests run two k-tiles ahead of their use: W by LDS-DMA into a ring of three slots, the lane's A floats into three register
    // sets.  Iteration t: wait until at most WAITN requests are in flight (= W of tile t and the floats of tile t + 1 have landed),
    // ONE barrier (tile t's W pieces of all four waves are visible; every wave is done with tile t - 1, whose ring slot the next DMA
    // reuses), request W of tile t + 2 and the floats of tile t + 3, then the MFMAs of tile t with the splitting of tile t + 1 in
    // their shadow.  Tile indices are clamped, not branched on: past the end the last tile is requested (and split) again, into a
    // slot / planes nobody reads; the final wait keeps the request registers alive until those loads have landed.
    // Period of the rotation: 3 register sets x 2 plane buffers = 6 tiles per trip of the loop.
    if (t_begin < t_end) {
        int wslot = 0, cslot = 0;
        auto next = [&](int sl) { return sl + 1 == 3 ? 0 : sl + 1; };
        request_w(t_begin, wslot); wslot = next(wslot);
        request_a(rA0, rA1, rA2, rA3);
        request_a(rB0, rB1, rB2, rB3);
        request_w(min(t_begin + 1, t_last), wslot); wslot = next(wslot);
        request_a(rC0, rC1, rC2, rC3);
        // the first tile's floats are split before the loop: nothing to hide them behind
        if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(10)" : "+v"(rA0), "+v"(rA1), "+v"(rA2), "+v"(rA3) : : "memory");
        else asm volatile("s_waitcnt vmcnt(12)" : "+v"(rA0), "+v"(rA1), "+v"(rA2), "+v"(rA3) : : "memory");
        gx_split8(rA0, rA1, ap0[0]);
        gx_split8(rA2, rA3, ap0[1]);
#define GX_STEP(I_, CUR_, NXT_, C0_, C1_, C2_, C3_, Q0_, Q1_, Q2_, Q3_)                                                          \
        if (t + I_ < t_end) {                                                                                                  \
            landed(C0_, C1_, C2_, C3_);                                                                                        \
            __syncthreads();                                                                                                   \
            request_w(min(t + I_ + 2, t_last), wslot); wslot = next(wslot);                                                    \
            request_a(Q0_, Q1_, Q2_, Q3_);                                                                                     \
            tile(cslot, CUR_, NXT_, C0_, C1_, C2_, C3_); cslot = next(cslot);                                                  \
        }
        for (int t = t_begin; t < t_end; t += 6) {
            GX_STEP(0, ap0, ap1, rB0, rB1, rB2, rB3, rA0, rA1, rA2, rA3)
            GX_STEP(1, ap1, ap0, rC0, rC1, rC2, rC3, rB0, rB1, rB2, rB3)
            GX_STEP(2, ap0, ap1, rA0, rA1, rA2, rA3, rC0, rC1, rC2, rC3)
            GX_STEP(3, ap1, ap0, rB0, rB1, rB2, rB3, rA0, rA1, rA2, rA3)
            GX_STEP(4, ap0, ap1, rC0, rC1, rC2, rC3, rB0, rB1, rB2, rB3)
            GX_STEP(5, ap1, ap0, rA0, rA1, rA2, rA3, rC0, rC1, rC2, rC3)
        }
#undef GX_STEP
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rA0), "+v"(rA1), "+v"(rA2), "+v"(rA3), "+v"(rB0), "+v"(rB1), "+v"(rB2), "+v"(rB3), "+v"(rC0),
                     "+v"(rC1), "+v"(rC2), "+v"(rC3) : : "memory");
    }
    __syncthreads();        // (no wave leaves -- and lets the workgroup's LDS be handed on -- while another wave's DMA may be in flight)

    // The product is computed transposed (W fragment as the MFMA's first operand): D[i][jm] with i = column of the group, jm = the
    // lane's own A row.  C/D layout of the 32x32 MFMA: jm = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- four consecutive
    // output columns per register quad: 16-byte stores, and the row's scale / residual belong to the lane that loaded the row.
    const bool mok = gm_a < M;
    float rs = 1.f;
    if (!slab && E.row_scale && mok) rs = E.row_scale[gm_a];
    float* dst = slab ? slab + ((size_t)by * Mcap + (mok ? gm_a : 0)) * N : C + (size_t)(mok ? gm_a : 0) * ldc;
    // (round 5) final results leave as full 128-byte lines through a 32 x 36-float patch of the (now free) ring that belongs to the
    // wave, as in gemm_x3r_kernel::finish_tile: sixteen bytes per lane into 64 different places per store instruction were a
    // request-rate problem next to the loads (r05 g1); K slices still go to their slab directly
    static_assert(3 * SB >= WAVES * 32 * 36 * 4, "the output patches must fit the ring");
    float* patch = (float*)Bs + wave * (32 * 36);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (!slab) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + 32 * j + 8 * q + 4 * (lane >> 5);
                float v[4] = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                if (mok && gn < N) {
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
                *(float4*)&patch[(lane & 31) * 36 + 8 * q + 4 * (lane >> 5)] = make_float4(v[0], v[1], v[2], v[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cc = n0 + 32 * j + 4 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (lane >> 3) + 8 * i;
                const float4 o = *(const float4*)&patch[r * 36 + 4 * (lane & 7)];
                const int gm = m0 + 32 * wave + r;
                if (gm < M && cc < N) *(float4*)&C[(size_t)gm * ldc + cc] = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
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

// =====================================================================================================================
// Resident-W, persistent form (round 5) for the tall contractions of the fine levels: M of 10^5 .. 10^6 rows, K <= 256, N <= 128.
// There the tile kernel above is all fixed cost: a 256 x 128 workgroup lives for 3 - 8 k-tiles, and with NO memory traffic at all
// (A rows aliased to the zero line, no stores) M = 707592, K = 96, N = 128 still takes 118 us of the 238 -- 2764 workgroups, one at
// a time per CU, each launching, staging its W slices through the ring, synchronising per k-tile and draining
// (profiles/r05_experiments.txt g1).  Here the whole pre-split W ([k-tile][column group][plane][32][40] bf16, <= 120 KB) is staged
// in LDS ONCE per CU and read in place; one 8-wave workgroup per CU; every WAVEFRONT walks its own sequence of 32-row tiles
// (tile = wave * gridDim.x + block, stride 8 gridDim.x: the chip's waves share the rows evenly at any M) with the same register
// pipeline as above (A floats straight from global memory in operand layout, three register sets, the splitting of k-tile t + 1 in
// the shadow of the MFMAs of k-tile t) -- and NO barrier after the staging: waves run out of phase and cover each other's
// latencies.  Per tile one full drain (the tile's first floats and the previous tile's stores together), counted waits inside.
#define GXR_PS 36                       // floats per row of a wave's output patch (32 + 4: the b128 accesses of a row group on distinct banks)
#define GXR_PATCH_BYTES (8 * 32 * GXR_PS * 4)
#define GXR_EPI_BYTES (2 * 128 * 4)          // per-column scale / shift of up to 128 columns, staged once
template <int TN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_x3r_kernel(const float* __restrict__ A, int lda, const unsigned short* __restrict__ Wx, int nkt, float* __restrict__ C, int ldc,
                int M, int N, GemmEpi E, const int* __restrict__ M_dev, GemmGather G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gxr_w[];      // [nkt][TN] chunks of GX_CHUNK bf16
    static_assert(TN == 1 || TN == 2 || TN == 4, "column groups per row tile");
    M = d3f_dyn(M, M_dev);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // ---- W: chunk (column group j, k-tile kt) of the packed copy -> chunk (kt, j) of the LDS image ----
        const int total = nkt * TN * 480;
        const uint4* src = (const uint4*)Wx;
        uint4* dst = (uint4*)gxr_w;
        for (int e = tid; e < total; e += 512) {
            const int c = e / 480, o = e - 480 * c;
            const int kt = c / TN, j = c - kt * TN;
            dst[e] = src[(size_t)(j * nkt + kt) * 480 + o];
        }
        // the per-column epilogue operands too (identity where absent): read from LDS in the epilogue -- a global load there is
        // waited for with vmcnt(0) by the compiler and drains the wave's prefetched operand loads once per tile
        float* ep = (float*)(gxr_w + (size_t)nkt * TN * (GX_CHUNK * 2) + GXR_PATCH_BYTES);
        for (int c = tid; c < 32 * TN; c += 512) {
            ep[c] = (E.col_scale && c < N) ? E.col_scale[c] : 1.f;
            ep[32 * TN + c] = (E.col_shift && c < N) ? E.col_shift[c] : 0.f;
        }
    }
    __syncthreads();
    const float* ep_scale = (const float*)(gxr_w + (size_t)nkt * TN * (GX_CHUNK * 2) + GXR_PATCH_BYTES);
    const float* ep_shift = ep_scale + 32 * TN;
    const int ntiles = (M + 31) >> 5;
    // 8 (lane >> 5), RECOMPUTED where it is used (two instructions, volatile: not kept live): at 256 registers the allocator spilled
    // this one value, and its reload -- a scratch load the compiler waits for with vmcnt(0) -- drained the wave's prefetched
    // operand loads at every tile boundary
    auto kofs_now = [&]() -> int {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (int)((l >> 5) << 3);
    };
    const int K1 = G.K1;
    const int n1 = G.gidx ? d3f_dyn(G.N1, G.N1_dev) : 0;
    const int tstride = 8 * (int)gridDim.x;
    const int rt0 = wave * (int)gridDim.x + (int)blockIdx.x;       // this wave's tiles: rt0, rt0 + tstride, ...
    if (rt0 >= ntiles) return;                                      // (after the workgroup's only barrier)
    const int ntl = (ntiles - rt0 + tstride - 1) / tstride;
    const int total = ntl * nkt;                                    // (tile, k-tile) pairs of this wave, walked as ONE stream
    // ---- request side: the rows of the tile whose floats are requested next (up to three k-tiles ahead of the multiplies, across
    // tile boundaries: a wave always has its next 12 KB on the way); a row that does not exist (beyond M, shadow / out-of-range
    // index) is the zero line, read at offset 0 whatever the k-tile ----
    int rq_tile = rt0, rq_kt = 0;
    const float* arow;
    const float* a2row;
    unsigned amask, a2mask;
    const float* pa;
    unsigned astep;
    auto a_rebase = [&](int t) {
        const bool first = t * GX_BK < K1;                 // (wave-uniform)
        const unsigned msk = first ? amask : a2mask;
        pa = (first ? arow : a2row) + ((unsigned)((first ? t * GX_BK : t * GX_BK - K1) + kofs_now()) & msk);
        astep = (unsigned)GX_BK & msk;
    };
    auto rq_rows = [&](int rt) {
        const int gm = rt * 32 + (lane & 31);
        arow = gd_zero_line; a2row = gd_zero_line; amask = 0u; a2mask = 0u;
        if (rt < ntiles && gm < M) {
            int sr = gm;
            if (G.gidx) {
                sr = G.gidx[(size_t)gm * G.ld_gidx];
                if (sr < 0 || sr >= n1) sr = -1;
            }
            if (sr >= 0) { arow = A + (size_t)sr * lda; amask = 0xffffffffu; }
            if (G.A2) { a2row = G.A2 + (size_t)gm * G.lda2; a2mask = 0xffffffffu; }
        }
        a_rebase(0);
    };
    rq_rows(rq_tile);
    gx_f4 rA0, rA1, rA2, rA3, rB0, rB1, rB2, rB3, rC0, rC1, rC2, rC3;
    uint4 ap0[2][3], ap1[2][3];
    auto request_a = [&](gx_f4& r0, gx_f4& r1, gx_f4& r2, gx_f4& r3) {
        gx_ld16(r0, pa);
        gx_ld16o<16>(r1, pa);
        gx_ld16o<64>(r2, pa);
        gx_ld16o<80>(r3, pa);
        // advance to the next (tile, k-tile) pair; past this wave's last tile: the zero line (rq_rows), nobody consumes it
        if (++rq_kt == nkt) {                                  // (wave-uniform)
            rq_kt = 0;
            rq_tile += tstride;
            rq_rows(rq_tile);
        } else if (rq_kt * GX_BK == K1) a_rebase(rq_kt);
        else pa += astep;
    };
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // one k-tile: the 12 TN MFMAs of k-tile kt on planes `cur`, the splitting of the NEXT pair's floats in their shadow (the
    // micro-operation list of gemm_x3_kernel::tile), W fragments read from the resident image
    auto tile = [&](int kt, const uint4 (&cur)[2][3], uint4 (&nxt)[2][3], const gx_f4& r0, const gx_f4& r1, const gx_f4& r2,
                    const gx_f4& r3) {
        const unsigned short* bp = (const unsigned short*)gxr_w + (size_t)kt * (TN * GX_CHUNK) + (lane & 31) * GX_LS + kofs_now();
        constexpr int NM = 12 * TN, NOPS = 88;
        constexpr int PA[2][6] = {{2, 1, 0, 1, 0, 0}, {0, 1, 0, 2, 1, 0}}, PB[2][6] = {{0, 1, 2, 0, 1, 0}, {2, 1, 1, 0, 0, 0}};
        uint4 b[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = *(const uint4*)(bp + j * GX_CHUNK + p * (32 * GX_LS));
        float x[16] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3], r3[0], r3[1], r3[2], r3[3]};
        float hl[16];
        unsigned P[3][8];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int s = m / (6 * TN), prod = (m % (6 * TN)) / TN, j = m % TN;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, b[j][PB[s][prod]]),
                                                             __builtin_bit_cast(gb_bf16x8, cur[s][PA[s][prod]]), acc[j], 0, 0, 0);
            if (s == 0 && (prod == 2 || prod >= 4)) {
                const int p = prod == 2 ? 2 : prod == 4 ? 1 : 0;
                b[j][p] = *(const uint4*)(bp + j * GX_CHUNK + p * (32 * GX_LS) + 16);
            }
#pragma unroll
            for (int k = m * NOPS / NM; k < (m + 1) * NOPS / NM; ++k) {
                const int lvl = k < 8 ? 0 : k < 24 ? 1 : k < 40 ? 2 : k < 48 ? 3 : k < 64 ? 4 : k < 80 ? 5 : 6;
                if (lvl == 0) P[0][k] = gx_cvt_pk(x[2 * k], x[2 * k + 1]);
                else if (lvl == 3) P[1][k - 40] = gx_cvt_pk(x[2 * (k - 40)], x[2 * (k - 40) + 1]);
                else if (lvl == 6) P[2][k - 80] = gx_cvt_pk(x[2 * (k - 80)], x[2 * (k - 80) + 1]);
                else if (lvl == 1 || lvl == 4) {
                    const int e = k - (lvl == 1 ? 8 : 48);
                    const unsigned pk = P[lvl == 1 ? 0 : 1][e >> 1];
                    hl[e] = __uint_as_float((e & 1) ? (pk & 0xffff0000u) : (pk << 16));
                } else {
                    const int e = k - (lvl == 2 ? 24 : 64);
                    x[e] = gx_sub(x[e], hl[e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) nxt[s][p] = make_uint4(P[p][4 * s], P[p][4 * s + 1], P[p][4 * s + 2], P[p][4 * s + 3]);
    };
    // ---- the finished tile of the compute side: epilogue and stores (D[i][jm] as in gemm_x3_kernel), accumulators cleared ----
    int c_tile = rt0, c_kt = 0;
    // Stores: in the accumulator layout a lane holds four consecutive columns of ITS row -- 16 bytes into each of 64 different
    // places per wave instruction, 22 M sixteen-byte write requests for the 362 MB of a level-0 layer (the stores alone cost 68 of
    // that layer's 238 us: g1).  The finished values of one 32-column group therefore go through a 32 x 36-float patch of LDS that
    // belongs to the wave (no barrier: a wave's LDS operations keep their order) and leave as FULL 128-byte lines: lane l writes
    // 16 bytes of row l / 8 + 8 i, eight lanes per row.
    float* patch = (float*)(gxr_w + (size_t)nkt * TN * (GX_CHUNK * 2)) + wave * (32 * GXR_PS);
    auto finish_tile = [&]() {
        const int gm_a = c_tile * 32 + (lane & 31);
        const bool mok = gm_a < M;
        float rs = 1.f;
        if (E.row_scale && mok) rs = E.row_scale[gm_a];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = 32 * j + 8 * q + 4 * (lane >> 5);
                float v[4] = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                if (mok && gn < N) {
                    const float4 c4 = *(const float4*)&ep_scale[gn];
                    const float4 h4 = *(const float4*)&ep_shift[gn];
                    const float4 r4 = E.residual ? *(const float4*)&E.residual[(size_t)gm_a * E.ldr + gn] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float c[4] = {c4.x, c4.y, c4.z, c4.w}, h[4] = {h4.x, h4.y, h4.z, h4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tt = (v[e] * rs) * c[e] + h[e];
                        tt += rr[e];
                        v[e] = (E.leaky && !(tt > 0.f)) ? tt * E.alpha : tt;
                    }
                }
                *(float4*)&patch[(lane & 31) * GXR_PS + 8 * q + 4 * (lane >> 5)] = make_float4(v[0], v[1], v[2], v[3]);
                acc[j][4 * q] = acc[j][4 * q + 1] = acc[j][4 * q + 2] = acc[j][4 * q + 3] = 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cc = 32 * j + 4 * (lane & 7);                  // this lane's four columns of the group
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (lane >> 3) + 8 * i;
                const float4 o = *(const float4*)&patch[r * GXR_PS + 4 * (lane & 7)];
                const int gm = c_tile * 32 + r;
                if (gm < M && cc < N) *(float4*)&C[(size_t)gm * ldc + cc] = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                         // (the patch is rewritten by the next column group)
        }
        c_tile += tstride;
    };
    // everything but the newest request (four loads) has retired.  Loads retire in order among themselves, so "at most four
    // operations outstanding" means every load older than the newest four has landed whatever stores are in flight beside them (a
    // store still in flight only makes the wait longer: the tile's first wait after its epilogue also waits for those stores).
    auto landed = [&](gx_f4& r0, gx_f4& r1, gx_f4& r2, gx_f4& r3) {
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
    };
    request_a(rA0, rA1, rA2, rA3);
    request_a(rB0, rB1, rB2, rB3);
    request_a(rC0, rC1, rC2, rC3);
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(rA0), "+v"(rA1), "+v"(rA2), "+v"(rA3) : : "memory");
    gx_split8(rA0, rA1, ap0[0]);
    gx_split8(rA2, rA3, ap0[1]);
#define GXR_STEP(I_, CUR_, NXT_, C0_, C1_, C2_, C3_, Q0_, Q1_, Q2_, Q3_)                                                         \
    if (it + I_ < total) {                                                                                                     \
        landed(C0_, C1_, C2_, C3_);                                                                                            \
        request_a(Q0_, Q1_, Q2_, Q3_);                                                                                         \
        tile(c_kt, CUR_, NXT_, C0_, C1_, C2_, C3_);                                                                            \
        if (++c_kt == nkt) { c_kt = 0; finish_tile(); }                                                                        \
    }
    for (int it = 0; it < total; it += 6) {
        GXR_STEP(0, ap0, ap1, rB0, rB1, rB2, rB3, rA0, rA1, rA2, rA3)
        GXR_STEP(1, ap1, ap0, rC0, rC1, rC2, rC3, rB0, rB1, rB2, rB3)
        GXR_STEP(2, ap0, ap1, rA0, rA1, rA2, rA3, rC0, rC1, rC2, rC3)
        GXR_STEP(3, ap1, ap0, rB0, rB1, rB2, rB3, rA0, rA1, rA2, rA3)
        GXR_STEP(4, ap0, ap1, rC0, rC1, rC2, rC3, rB0, rB1, rB2, rB3)
        GXR_STEP(5, ap1, ap0, rA0, rA1, rA2, rA3, rC0, rC1, rC2, rC3)
    }
#undef GXR_STEP
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rA0), "+v"(rA1), "+v"(rA2), "+v"(rA3), "+v"(rB0), "+v"(rB1), "+v"(rB2), "+v"(rB3), "+v"(rC0),
                 "+v"(rC1), "+v"(rC2), "+v"(rC3) : : "memory");
}
