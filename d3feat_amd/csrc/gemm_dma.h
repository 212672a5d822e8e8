// fp32 contraction with LDS-DMA staging (round 4): the operands travel global -> LDS by `global_load_lds_dwordx4`, no staging
// registers and no ds_write pass, which frees the register file for a STAGES-deep ring of k-tiles in flight per workgroup
// (round 3's register-staged kernel could prefetch exactly one tile: a k-step lasted about one loaded-memory round trip).
//
//   * Both operands are "row = output index, k contiguous": A f32[M, K] as the features are stored; the weights PRE-TRANSPOSED
//     once per tensor to Wt f32[N][Kp] (Kp = K rounded up to 32, zero padded; d3f_gemm_pack_f32t) -- a DMA piece copies 16
//     contiguous bytes per lane, so it cannot transpose.
//   * LDS image of a stage: (BM + BN) rows x 32 floats, 128-byte rows, NO padding (a DMA wave-instruction writes 64 lanes x 16 B
//     = 8 whole rows linearly).  Bank conflicts of the fragment reads are removed by an XOR swizzle of the 16-byte chunk index
//     with bits 1..3 of the row, applied on BOTH sides: the lane that fills LDS slot (row, c') fetches global chunk
//     c' ^ ((row >> 1) & 7), the fragment read of chunk c goes to slot c ^ ((row >> 1) & 7).  For the lane groups of ds_read_b128
//     ({0-3, 12-15, 20-27}, ...) the 16 (row & 1, chunk ^ swizzle) pairs are distinct = 64 distinct banks.
//   * Rows that must read as zero (beyond M / N, shadow rows of the gathered operand, k beyond K of the un-padded operand A) take
//     their bytes from a 128-byte zero line in global memory: a DMA lane that is switched off would leave stale LDS bytes.
//   * One s_barrier per k-step.  Step t: wait until only the newer tiles' pieces are outstanding (counted s_waitcnt vmcnt, never
//     through __syncthreads, which would drain the whole queue) -> barrier (every wave's pieces of tile t have landed, every wave
//     has finished reading tile t - 1) -> issue tile t + STAGES - 1 into the slot tile t - 1 occupied -> fragments + MFMAs of
//     tile t.
// Same k-permuted fragments (lane (r, h) owns k = 16 h .. 16 h + 15 of its row: four ds_read_b128 per operand and k-tile), same
// transposed accumulators / epilogue as gemm_fast_kernel.
#pragma once

#define GD_BK 32
__device__ __attribute__((aligned(128))) float gd_zero_line[32];     // zero-initialised device global

typedef const __attribute__((address_space(1))) void* gd_gptr;
typedef __attribute__((address_space(3))) void* gd_lptr;

template <int N_> __device__ __forceinline__ void gd_wait_vm() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N_ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N_ == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N_ == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N_ == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else static_assert(N_ < 0, "add the count");
}

// WM x WN wavefronts, TM x TN 32 x 32 accumulator tiles each (independent MFMA chains; an A / B fragment feeds TN / TM MFMAs),
// STAGES k-tiles of LDS; EPI bit 0: per-row scale, bit 1: residual operand.
//
// PERSISTENT over work items.  An item = (row block, K slice, column tile); the grid holds G workgroups (G a multiple of 8, at most
// what is resident at once), workgroup w takes the items w, w + G, w + 2 G, ... of the XCD-aware item order (common.h:
// d3f_xcd_tile -- the column tiles of a row block, which read the same A rows, meet in one L2).  The DMA ring does not stop at an
// item boundary: while the last k-tiles of item r are multiplied and its output tile is stored, the first k-tiles of item r + 1
// are already landing.  A workgroup of the one-item-per-workgroup form lived for 2 - 16 k-steps and paid a dispatch gap, a cold
// first load (one full memory round trip with nothing to overlap it) and a store tail each time: ~30 % of its life at the
// network's shapes (the bare LDS + MFMA loop runs at 0.90 of the matrix peak in isolation, tools/ubench/mfma_rate.hip, and reached
// 0.70 inside such workgroups: profiles/r03_experiments.txt x9).
template <int WM, int WN, int TM, int TN, int STAGES, int EPI>
__global__ void __launch_bounds__(256)
gemm_dma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int Kp, float* __restrict__ C, int ldc,
                int M, int N, int K, int tiles_per_split, int S, float* __restrict__ slab, GemmEpi E,
                const int* __restrict__ M_dev, GemmGather G) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, ROWS_T = BM + BN;
    constexpr int P = ROWS_T / 32;                 // DMA pieces per thread and stage (256 threads x 16 B = 32 rows per piece)
    constexpr int PA = BM / 32;                    // the first PA pieces are A rows, the rest weight rows
    constexpr int SF = ROWS_T * GD_BK;             // floats per stage
    static_assert(WM * WN == 4 && STAGES >= 2 && STAGES <= 3, "tile shape");
    constexpr bool ROWS = (EPI & 1) != 0, RES = (EPI & 2) != 0;
    const int Mcap = M;
    M = d3f_dyn(M, M_dev);
    const unsigned ncol = (unsigned)((N + BN - 1) / BN), gxy = ncol * (unsigned)S;
    const unsigned items = gxy * (unsigned)((M + BM - 1) / BM);
    const unsigned Gw = gridDim.x;
    if (blockIdx.x >= items) return;
    const int R = (int)((items - blockIdx.x + Gw - 1) / Gw);      // items of this workgroup
    extern __shared__ __attribute__((aligned(128))) float gd_smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nt_all = (K + GD_BK - 1) / GD_BK;
    const int kend1 = G.A2 ? G.K1 : K;             // columns [0, kend1) come from A, [kend1, K) from the second operand
    const int n1 = d3f_dyn(G.N1, G.N1_dev);
    const int r8 = tid >> 3;                       // row inside a piece

    // ---- the DMA cursor: (item, k-tile) of the next tile to request, with this thread's sources for that item.
    // piece p fills LDS slot (row = 32 p + tid / 8, chunk position tid % 8) of a stage
    const float* s1[P];           // A row (gathered or in place) / weight row, advanced to this thread's chunk
    const float* s2[PA];          // second-operand row, advanced to the chunk and rebased to k = 0
    bool ok1[P], ok2[PA];
    int c_r = 0, c_t = 0, c_tend = 0;
    auto item_coords = [&](int r, unsigned& bz, unsigned& by, unsigned& bx) {
        const unsigned T_ = d3f_xcd_tile(blockIdx.x + (unsigned)r * Gw, items);
        bz = T_ / gxy; by = (T_ % gxy) / ncol; bx = T_ % ncol;
    };
    auto cursor_setup = [&](int r) {
        unsigned bz, by, bx;
        item_coords(r, bz, by, bx);
        const int m0 = bz * BM, n0 = bx * BN;
        c_t = by * tiles_per_split;
        c_tend = min(nt_all, c_t + tiles_per_split);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int row = 32 * p + r8;
            const int kc = ((tid & 7) ^ ((row >> 1) & 7)) << 2;          // k offset of the global chunk this slot holds
            if (p < PA) {
                const int gm = m0 + row;
                const bool in = gm < M;
                int sr = in ? gm : 0;
                bool ok = in;
                if (G.gidx) {
                    sr = G.gidx[(size_t)(in ? gm : 0) * G.ld_gidx];
                    ok = in && sr >= 0 && sr < n1;                       // shadow neighbour: zero row
                    sr = ok ? sr : 0;
                }
                s1[p] = A + (size_t)sr * lda + kc;
                ok1[p] = ok;
                s2[p < PA ? p : 0] = G.A2 ? G.A2 + (size_t)(in ? gm : 0) * G.lda2 + kc - kend1 : A;
                ok2[p < PA ? p : 0] = in && G.A2 != nullptr;
            } else {
                const int gn = n0 + row - BM;
                ok1[p] = gn < N;
                s1[p] = Wt + (size_t)(ok1[p] ? gn : 0) * Kp + kc;
            }
        }
    };
    // LDS byte address of this wave's 1 KiB of piece 0, stage 0 (wave-uniform -> SGPR)
    const unsigned lds_wave = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((unsigned)(unsigned long long)(gd_lptr)gd_smem + (unsigned)(tid >> 6) * 1024u));
    auto issue_tile = [&](int t, int slot) {
        const unsigned dst = lds_wave + (unsigned)slot * (unsigned)(SF * 4);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float* src;
            if (p < PA) {
                const int row = 32 * p + r8;
                const int k = t * GD_BK + ((((tid & 7) ^ ((row >> 1) & 7))) << 2);
                const bool second = k >= kend1;
                const bool ok = (second ? ok2[p < PA ? p : 0] : ok1[p]) && k < K;
                src = second ? s2[p < PA ? p : 0] + t * GD_BK : s1[p] + t * GD_BK;
                src = ok ? src : gd_zero_line;
            } else {
                src = ok1[p] ? s1[p] + t * GD_BK : gd_zero_line;
            }
            // LDS-DMA through inline asm: hipcc would count a builtin DMA as a store to LDS and drain the whole queue
            // (s_waitcnt vmcnt(0)) before the next ds_read -- which is exactly the prefetch this kernel exists for.  An asm
            // statement is absent from its bookkeeping; its completion is counted by hand (gd_wait_vm below).
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst + (unsigned)(p * 4096)) : "memory");
        }
    };
    int slot_w = 0, ahead = 0;                    // next stage to fill; tiles requested and not yet consumed
    auto issue_next = [&]() {
        if (c_r >= R) return;
        issue_tile(c_t, slot_w);
        slot_w = slot_w + 1 == STAGES ? 0 : slot_w + 1;
        ++ahead;
        if (++c_t == c_tend) {
            if (++c_r < R) cursor_setup(c_r);
        }
    };

    // fragment read offsets (floats, inside a stage): row R, chunk (4 h + q) ^ ((R >> 1) & 7)
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    const int arow_off = (wm * TM * 32 + fr) * GD_BK, brow_off = (BM + wn * TN * 32 + fr) * GD_BK;
    int coff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) coff[q] = (((4 * fh + q) ^ fsw) << 2);

    cursor_setup(0);
#pragma unroll
    for (int d = 0; d < STAGES - 1; ++d) issue_next();
    int slot = 0;
    for (int r = 0; r < R; ++r) {
        unsigned bz, by, bx;
        item_coords(r, bz, by, bx);
        const int m0 = bz * BM, n0 = bx * BN;
        const int t_begin = by * tiles_per_split;
        const int nt = min(nt_all, t_begin + tiles_per_split) - t_begin;

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        for (int i = 0; i < nt; ++i) {
            // `ahead` tiles are in flight and the oldest is the one multiplied now; the newer ones may stay outstanding --
            // except on the first step of an item: the previous item's output stores are still in the queue then (stores and
            // loads share the counter and complete out of order with respect to each other), so everything is awaited once
            if (ahead <= 1 || i == 0) gd_wait_vm<0>();
            else gd_wait_vm<P>();
            __builtin_amdgcn_s_barrier();
            --ahead;
            issue_next();
            const float* as = gd_smem + slot * SF + arow_off;
            const float* bs = gd_smem + slot * SF + brow_off;
            float4 fa[TM][4], fb[TN][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int ii = 0; ii < TM; ++ii) fa[ii][q] = *(const float4*)&as[ii * 32 * GD_BK + coff[q]];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j][q] = *(const float4*)&bs[j * 32 * GD_BK + coff[q]];
            }
            __builtin_amdgcn_sched_barrier(0);        // every fragment read is issued before the first MFMA waits on one
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int ii = 0; ii < TM; ++ii) {
                        const float a = e == 0 ? fa[ii][q].x : e == 1 ? fa[ii][q].y : e == 2 ? fa[ii][q].z : fa[ii][q].w;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float b = e == 0 ? fb[j][q].x : e == 1 ? fb[j][q].y : e == 2 ? fb[j][q].z : fb[j][q].w;
                            // operands swapped: the accumulator holds the TRANSPOSED tile (a lane owns one output row, four
                            // consecutive columns per register quad -> 16-byte stores)
                            acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[ii][j], 0, 0, 0);
                        }
                    }
                }
            }
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }

        // ---- epilogue of the item: as gemm_fast_kernel's (per-column terms up front, no load between two stores).  Its ordinary
        // loads make hipcc wait for vmcnt(0), i.e. also for the next item's first tiles, which are needed one step later anyway.
        float4 cs4[TN][4], ch4[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + (wn * TN + j) * 32 + 8 * q + 4 * (lane >> 5);
                const bool nok = gn < N;
                cs4[j][q] = (!slab && E.col_scale && nok) ? *(const float4*)&E.col_scale[gn] : make_float4(1.f, 1.f, 1.f, 1.f);
                ch4[j][q] = (!slab && E.col_shift && nok) ? *(const float4*)&E.col_shift[gn] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + (wm * TM + i) * 32 + (lane & 31);
            const bool mok = gm < M;
            float rs = 1.f;
            if (ROWS && !slab) rs = E.row_scale[mok ? gm : M - 1];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 o[4], res[4];
                if (RES && !slab) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gn = n0 + (wn * TN + j) * 32 + 8 * q + 4 * (lane >> 5);
                        res[q] = (mok && gn < N) ? *(const float4*)&E.residual[(size_t)gm * E.ldr + gn] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    if (!slab) {
                        const float c[4] = {cs4[j][q].x, cs4[j][q].y, cs4[j][q].z, cs4[j][q].w};
                        const float h4[4] = {ch4[j][q].x, ch4[j][q].y, ch4[j][q].z, ch4[j][q].w};
                        const float r4[4] = {RES ? res[q].x : 0.f, RES ? res[q].y : 0.f, RES ? res[q].z : 0.f, RES ? res[q].w : 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = (ROWS ? v[e] * rs : v[e]) * c[e] + h4[e];
                            if (RES) t += r4[e];
                            v[e] = (E.leaky && !(t > 0.f)) ? t * E.alpha : t;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(v[e]));
                    o[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
                float* dst = slab ? slab + ((size_t)by * Mcap + (mok ? gm : 0)) * N : C + (size_t)(mok ? gm : 0) * ldc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gn = n0 + (wn * TN + j) * 32 + 8 * q + 4 * (lane >> 5);
                    if (mok && gn < N) *(float4*)&dst[gn] = o[q];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) gemm_pack_f32t_kernel(const float* __restrict__ B, int ldb, int K, int N, int Kp,
                                                            float* __restrict__ Wt) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * Kp) return;
    const int n = (int)(t / Kp), k = (int)(t % Kp);
    Wt[t] = k < K ? B[(size_t)k * ldb + n] : 0.f;
}
