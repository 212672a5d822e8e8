// fp32 GEMM on the gfx950 matrix cores with the D3Feat inference epilogue fused.
//
// Reference ops replaced: kernels/convolution_ops.py:90-99 (unary_convolution = tf.matmul), :243-253 (KPConv
// kernel-weight contraction, sum over kernel points, neighbour-count normalisation), followed by
// models/network_blocks.py:149-160 (batch_norm, inference) and :185-186 (leaky_relu) and the residual add of
// the resnet blocks (:368, :612):
//   C[m,n] = act( acc[m,n] * row_scale[m] * col_scale[n] + col_shift[n] + residual[m,n] ),  acc = A @ B
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, 64 cycles per SIMD, exact fp32 products and sums) -- the 1e-4 parity bar
// is an fp32 bar, so the contraction stays in fp32 on the MFMA pipe (157 TF peak).
// Two kernels, one launcher (gemm_run):
//   gemm_fast_kernel    the production tile kernel for float4-addressable operands (every shape of the network): 256-thread
//                       workgroup = 4 wavefronts x (TM x TN) 32x32 accumulator tiles, BK = 32, double-buffered LDS, k-permuted
//                       fragments read with ds_read_b128, transposed accumulators -> 16-byte stores, straight-line staging;
//                       workgroup tile 64x64 (default) or 128x32 (Cout <= 32).  Measured and removed again: 128x64 / 128x128 register
//                       tiles (slower on every shape of the network: tails), a streaming form for many rows x shallow K (B slab
//                       resident in LDS, A global -> registers: equal end to end), a second prefetch register set (round 3);
//   gemm_f32_kernel     generic fallback (odd K / N / leading dimensions, unaligned bases): scalar tail handling.
// Skinny problems (few output tiles, long K: the deep KPConv layers) are split along K into slabs that a second kernel
// reduces in a fixed order (deterministic).
#include "common.h"
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GM_BK 32
#define GM_SA (GM_BK + 1)

// Optional composite A operand (decoder of models/D3Feat.py:55-63): A = [ x'[gidx[m, 0]] | A2[m] ] -- the nearest-upsample
// gather (closest_pool, models/network_blocks.py:69-83: x' = x + zero row) and the skip concatenation feed the unary
// contraction directly, so the concatenated [N, C1 + C2] tensor is never written to / re-read from HBM.
struct GemmGather {
    const int* gidx;      // NULL: A rows are used in place
    int ld_gidx;
    int N1;               // rows of A (the gather source); indices outside [0, N1) read the zero row
    const int* N1_dev;
    const float* A2;      // NULL: no second operand
    int lda2;
    int K1;               // columns taken from A (multiple of 4 when A2 != NULL)
};

struct GemmEpi {
    const float* row_scale;
    const float* col_scale;
    const float* col_shift;
    const float* residual;
    int ldr;
    int leaky;
    float alpha;
};

__device__ __forceinline__ float gemm_epilogue(float v, int m, int n, const GemmEpi& E) {
    if (E.row_scale) v *= E.row_scale[m];
    if (E.col_scale) v *= E.col_scale[n];
    if (E.col_shift) v += E.col_shift[n];
    if (E.residual) v += E.residual[(size_t)m * E.ldr + n];
    if (E.leaky) v = v > 0.f ? v : v * E.alpha;
    return v;
}

template <int WM, int WN, int TM, int TN>  // waves along M / N (WM*WN == 4); 32x32 MFMA tiles per wave along M / N
__global__ void __launch_bounds__(256)
gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                int M, int N, int K, int vecA, int vecB, int tiles_per_split, float* __restrict__ slab, GemmEpi E,
                const int* __restrict__ M_dev, GemmGather G) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int A_F4 = BM * GM_BK / 4 / 256;  // float4 loads per thread for the A tile
    constexpr int B_F4 = GM_BK * BN / 4 / 256;  // ... for the B tile (>= 1)
    static_assert(WM * WN == 4 && A_F4 >= 1 && B_F4 >= 1, "tile shape");
    const int Mcap = M;                       // slab stride stays the capacity
    M = d3f_dyn(M, M_dev);
    // grid = (n tiles, k splits, m tiles): the row tile is the SLOWEST dispatch dimension, so in capacity mode the live
    // workgroups are the first contiguous run of the dispatch order (dense over XCDs / CUs) and the empty ones trail
    if ((int)(blockIdx.z * BM) >= M) return;  // capacity-sized grid: row block beyond the real row count
    __shared__ float As[2][BM * GM_SA];                                 // double buffered: one barrier per k-tile
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.z * BM, n0 = blockIdx.x * BN;  // neighbouring workgroups share the A row tile in L2
    const int nt_all = (K + GM_BK - 1) / GM_BK;
    const int t_begin = blockIdx.y * tiles_per_split;
    const int t_end = min(nt_all, t_begin + tiles_per_split);

    float4 ra[A_F4], rb[B_F4];

    // source row of each A-tile row this thread stages (fixed across k-tiles): the row itself, or the gathered one
    int srow[A_F4];
    {
        const int n1 = d3f_dyn(G.N1, G.N1_dev);
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int gm = m0 + ((tid + i * 256) >> 3);
            int sr = gm < M ? gm : -1;
            if (G.gidx && sr >= 0) {
                sr = G.gidx[(size_t)gm * G.ld_gidx];
                if (sr < 0 || sr >= n1) sr = -2;      // shadow neighbour: zero row
            }
            srow[i] = sr;
        }
    }

    auto load_tile = [&](int t) {
        const int k0 = t * GM_BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;          // float4 slot: row = e / 8, k4 = e % 8
            const int r = e >> 3, k = k0 + ((e & 7) << 2);
            const int gm = m0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                if (G.A2 && k >= G.K1) {          // second operand: skip features, row gm
                    const float* p = G.A2 + (size_t)gm * G.lda2 + (k - G.K1);
                    if (k + 3 < K) v = *(const float4*)p;
                    else {
                        if (k < K) v.x = p[0];
                        if (k + 1 < K) v.y = p[1];
                        if (k + 2 < K) v.z = p[2];
                        if (k + 3 < K) v.w = p[3];
                    }
                } else if (srow[i] >= 0) {
                    const int kend = G.A2 ? G.K1 : K;
                    const float* p = A + (size_t)srow[i] * lda + k;
                    if (vecA && k + 3 < kend) v = *(const float4*)p;
                    else {
                        if (k < kend) v.x = p[0];
                        if (k + 1 < kend) v.y = p[1];
                        if (k + 2 < kend) v.z = p[2];
                        if (k + 3 < kend) v.w = p[3];
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;          // row = e / (BN/4), n4 = e % (BN/4)
            const int r = e / (BN / 4), n = n0 + ((e % (BN / 4)) << 2);
            const int gk = k0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                const float* p = B + (size_t)gk * ldb + n;
                if (vecB && n + 3 < N) v = *(const float4*)p;
                else {
                    if (n < N) v.x = p[0];
                    if (n + 1 < N) v.y = p[1];
                    if (n + 2 < N) v.z = p[2];
                    if (n + 3 < N) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;
            const int r = e >> 3, k = (e & 7) << 2;
            float* d = &As[buf][r * GM_SA + k];
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;
            const int r = e / (BN / 4), n = (e % (BN / 4)) << 2;
            *(float4*)&Bs[buf][r * BN + n] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
    }
    __syncthreads();
    // A fragment: lane reads column (k + lane/32) of row (lane%32) of its 32-row tile; B fragment: row (k + lane/32), col lane%32
    const int arow = (wm * TM * 32 + (lane & 31)) * GM_SA + (lane >> 5);
    const int bcol = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) load_tile(t + 1);   // global -> registers, in flight while this tile is multiplied
#pragma unroll
        for (int kk = 0; kk < GM_BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][arow + i * 32 * GM_SA + kk * 2];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][bcol + j * 32 + kk * 2 * BN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < t_end) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm < M && gn < N) {
                    if (slab) slab[((size_t)blockIdx.y * Mcap + gm) * N + gn] = acc[i][j][r];
                    else C[(size_t)gm * ldc + gn] = gemm_epilogue(acc[i][j][r], gm, gn, E);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The production tile kernel: same tiling as gemm_f32_kernel above, written so that the compiler has nothing to serialise.
//  * operands are float4-addressable (K, N, leading dimensions multiples of 4, 16-byte aligned bases: every shape of the
//    network), so a tile load is straight-line code: clamped address, one global_load_dwordx4, a select to zero -- no
//    branch between the loads and therefore no s_waitcnt before the multiply they are meant to overlap with;
//  * the contraction may visit k in any order as long as A and B agree: within a 32-deep tile lane (r, h) owns
//    k = 16h .. 16h+15, which are CONSECUTIVE in an LDS row, so a wavefront fetches its A and B fragments for the whole
//    tile with four ds_read_b128 each (B is stored transposed, [n][k]) instead of 32 ds_read_b32 followed by a wait apiece;
//    the second half of the fragments lands while the first eight MFMAs run.
// Row stride 36 floats: 16-byte aligned rows, and 36r mod 64 spreads eight consecutive rows over all banks for b128 reads.
// ------------------------------------------------------------------------------------------------
#define GF_S 36

// NACC: accumulator chains per MFMA tile; EPI bit 0: per-row scale, bit 1: residual operand
template <int WM, int WN, int TM, int TN, int NACC, int EPI>
__global__ void __launch_bounds__(256)
gemm_fast_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                 int M, int N, int K, int tiles_per_split, float* __restrict__ slab, GemmEpi E,
                 const int* __restrict__ M_dev, GemmGather G) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int A_F4 = BM * GM_BK / 4 / 256, B_F4 = GM_BK * BN / 4 / 256;
    static_assert(WM * WN == 4 && A_F4 >= 1 && B_F4 >= 1, "tile shape");
    const int Mcap = M;
    constexpr bool ROWS = (EPI & 1) != 0, RES = (EPI & 2) != 0;
    M = d3f_dyn(M, M_dev);
    if ((int)(blockIdx.z * BM) >= M) return;
    // live workgroups = the first contiguous run of the dispatch order (row tile is the slowest grid dimension); among them
    // every XCD takes one contiguous run of (row tile, K slice, column tile) triples, so the column tiles of a row block --
    // which read the same A rows -- meet in one L2 (common.h: d3f_xcd_tile)
    const unsigned gx_ = gridDim.x, gxy_ = gridDim.x * gridDim.y;
    const unsigned T_ = d3f_xcd_tile(blockIdx.x + gx_ * blockIdx.y + gxy_ * blockIdx.z, gxy_ * (unsigned)((M + BM - 1) / BM));
    const unsigned bz = T_ / gxy_, by = (T_ % gxy_) / gx_, bx = T_ % gx_;
    extern __shared__ __attribute__((aligned(16))) float gf_smem[];
    float* As = gf_smem;                          // [2][BM][GF_S]
    float* Bt = gf_smem + 2 * BM * GF_S;          // [2][BN][GF_S]   B tile transposed: Bt[n][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = bz * BM, n0 = bx * BN;
    const int nt_all = (K + GM_BK - 1) / GM_BK;
    const int t_begin = by * tiles_per_split;
    const int t_end = min(nt_all, t_begin + tiles_per_split);

    // per-thread staging slots: A slot i = row (tid + 256 i) / 8, k offset 4 * ((tid + 256 i) % 8)
    const float* arow[A_F4];      // source row in A (gathered or in place); A itself when the row reads as zero
    const float* a2row[A_F4];     // source row in the second operand
    bool aok[A_F4], a2ok[A_F4];
    const int ak = (tid & 7) << 2;
    {
        const int n1 = d3f_dyn(G.N1, G.N1_dev);
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int gm = m0 + ((tid + i * 256) >> 3);
            const bool in = gm < M;
            int sr = in ? gm : 0;
            bool ok = in;
            if (G.gidx) {
                sr = G.gidx[(size_t)(in ? gm : 0) * G.ld_gidx];
                ok = in && sr >= 0 && sr < n1;       // shadow neighbour: zero row
                sr = ok ? sr : 0;
            }
            arow[i] = A + (size_t)sr * lda;
            aok[i] = ok;
            a2row[i] = G.A2 ? G.A2 + (size_t)(in ? gm : 0) * G.lda2 : A;
            a2ok[i] = in && G.A2 != nullptr;
        }
    }
    const int kend1 = G.A2 ? G.K1 : K;             // columns [0, kend1) come from A, [kend1, K) from the second operand
    // B slot i of this thread: 16-row unit u = 16 i + tid / 16 of the [2][BN / 4] grid of (k half, float4 column) units,
    // row tid % 16 of it.  A 32-lane store group then covers 16 consecutive k of two neighbouring float4 columns, which
    // the transposed LDS rows (stride 36 = 4 mod 32) map to 32 different banks.
    int bk[B_F4], bn[B_F4];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int u = 16 * i + (tid >> 4);
        bk[i] = 16 * (u / (BN / 4)) + (tid & 15);
        bn[i] = (u % (BN / 4)) << 2;
    }

    float4 ra[A_F4], rb[B_F4];
    auto load_tile = [&](int t) {
        const int k = t * GM_BK + ak;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const bool second = k >= kend1;
            const bool ok = (second ? a2ok[i] : aok[i]) && k < K;
            const float* p = second ? a2row[i] + (k - kend1) : arow[i] + k;
            const float4 v = *(const float4*)(ok ? p : A);
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int gk = t * GM_BK + bk[i];
            const bool ok = n0 + bn[i] < N && gk < K;
            const float4 v = *(const float4*)(ok ? B + (size_t)gk * ldb + n0 + bn[i] : B);
            rb[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * GF_S;
        float* bt = Bt + buf * BN * GF_S;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *(float4*)&as[((tid + i * 256) >> 3) * GF_S + ak] = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            float* d = &bt[bn[i] * GF_S + bk[i]];
            d[0] = rb[i].x; d[GF_S] = rb[i].y; d[2 * GF_S] = rb[i].z; d[3 * GF_S] = rb[i].w;
        }
    };

    f32x16 acc[TM][TN][NACC];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.f;

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
    }
    __syncthreads();
    const int afrag = (wm * TM * 32 + (lane & 31)) * GF_S + (lane >> 5) * 16;
    const int bfrag = (wn * TN * 32 + (lane & 31)) * GF_S + (lane >> 5) * 16;
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);               // global -> registers, in flight while this tile is multiplied
        const float* as = As + cur * BM * GF_S + afrag;
        const float* bt = Bt + cur * BN * GF_S + bfrag;
        float4 fa[TM][4], fb[TN][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][q] = *(const float4*)&as[i * 32 * GF_S + 4 * q];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][q] = *(const float4*)&bt[j * 32 * GF_S + 4 * q];
        }
        __builtin_amdgcn_sched_barrier(0);        // every fragment read is issued before the first MFMA waits on one
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float a = e == 0 ? fa[i][q].x : e == 1 ? fa[i][q].y : e == 2 ? fa[i][q].z : fa[i][q].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float b = e == 0 ? fb[j][q].x : e == 1 ? fb[j][q].y : e == 2 ? fb[j][q].z : fb[j][q].w;
                        // operands swapped: the accumulator holds the TRANSPOSED tile, i.e. a lane owns one output row
                        // and four consecutive columns per register quad -> 16-byte stores in the epilogue
                        acc[i][j][e % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i][j][e % NACC], 0, 0, 0);
                    }
                }
            }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // Accumulator layout (operands swapped, so D = tile^T): lane l owns output row (l & 31) of its 32-row tile and, in
    // register quad q, the four consecutive columns 8q + 4(l >> 5) .. +3 of the 32-column tile: one global_store_dwordx4 per
    // quad (the store path is issue bound: 4 wide stores per tile instead of 16 dword stores).
    // No load may sit between two stores: a conditional load inside the store loop makes the compiler drain the memory
    // counter (s_waitcnt vmcnt(0)) before every store, i.e. one HBM round trip per element.  So the per-column terms are
    // fetched up front, the per-row scale (ROWS: the KPConv neighbour-count division) once per tile row, every output value
    // is finished (and pinned) before the first predicated store; a residual operand (RES: the identity shortcut of the
    // strided resnet blocks) is fetched as the four 16-byte pieces of a tile row before that tile's stores.
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
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][0][4 * q + e];
                    if (NACC == 2) v[e] += acc[i][j][NACC - 1][4 * q + e];
                }
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

#include "gemm_dma.h"

// res_bf16 / c_bf16: the residual operand / the output hold bfloat16 values (bf16 feature storage, d3f_gemm_bf16)
__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(const float* __restrict__ slab, int S, int M, int N, float* __restrict__ C, int ldc, GemmEpi E,
                          const int* __restrict__ M_dev, int res_bf16 = 0, int c_bf16 = 0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)d3f_dyn(M, M_dev) * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += slab[((size_t)s * M + m) * N + n];
    if (res_bf16 && E.residual) {
        const float r = d3f_bf16_f32(((const unsigned short*)E.residual)[(size_t)m * E.ldr + n]);
        GemmEpi E2 = E;
        E2.residual = nullptr;
        E2.leaky = 0;
        v = gemm_epilogue(v, m, n, E2) + r;
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
    } else {
        v = gemm_epilogue(v, m, n, E);
    }
    if (c_bf16) ((unsigned short*)C)[(size_t)m * ldc + n] = (unsigned short)d3f_bf16_rne(v);
    else C[(size_t)m * ldc + n] = v;
}

// Tile selection.  Large tiles (each wave owns 2x2 / 2x1 MFMA tiles: half the LDS traffic per flop, 4 independent
// accumulator chains) when the problem still fills the chip with them; smaller tiles / split K for the skinny deep layers.
static void gemm_plan(int M, int N, int K, int M_hint, int& bm, int& bn, int& S, int& tps) {
    // capacity mode: M is only an upper bound; split K for the row count the caller EXPECTS (skinny deep layers would
    // otherwise be planned as if they filled the chip and run their whole K loop in a handful of workgroups)
    if (M_hint > 0 && M_hint < M) M = M_hint;
    // Measured on MI355X over the network's 37 shapes (round 1 / 2 sweeps): these GEMMs are small (<= 7 GFLOP) and latency /
    // bandwidth bound, so the 64x64 tile -- 37 KB of LDS, 4 workgroups resident per CU -- beat the register-tiled 128x128 /
    // 128x64 variants everywhere.
    auto blocks_of = [&](int m, int n) { return (long long)d3f_cdiv(M, m) * d3f_cdiv(N, n); };
    if (N <= 32) { bm = 128; bn = 32; }
    else { bm = 64; bn = 64; }
    const long long blocks = blocks_of(bm, bn);
    const int nt = d3f_cdiv(K, GM_BK);
    // Skinny problems with a long K are latency bound per k-tile (global -> LDS -> MFMA): give every CU ~6 co-resident
    // workgroups by splitting K, as long as each split keeps >= 8 k-tiles.  Up to 16 k-tiles (K <= 512) a split never paid
    // for its slab traffic and reduce launch (tools/gemm_bench.py sweep).
    S = 1;
    if (blocks < 768 && nt > 16) {
        long long want = (1536 + blocks - 1) / blocks;
        long long maxs = nt / 8;
        S = (int)(want < maxs ? want : maxs);
        if (S > 64) S = 64;
        if (S < 1) S = 1;
    }
    tps = d3f_cdiv(nt, S);
    S = d3f_cdiv(nt, tps);
}

extern "C" size_t d3f_gemm_workspace_bytes(int M, int N, int K, int M_hint) {
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    int bm, bn, S, tps;
    gemm_plan(M, N, K, M_hint, bm, bn, S, tps);
    return S > 1 ? d3f_align((size_t)S * M * N * sizeof(float)) + 256 : 256;
}

static int gemm_run(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, GemmEpi E,
                    GemmGather G, void* workspace, size_t workspace_bytes, const int* M_dev, int M_hint, hipStream_t stream);

extern "C" int d3f_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                            const float* row_scale, const float* col_scale, const float* col_shift,
                            const float* residual, int ldr, int leaky, float alpha,
                            void* workspace, size_t workspace_bytes, const int* M_dev, int M_hint, void* stream_) {
    if (M < 0 || N < 0 || K < 0 || lda < K || ldb < N || ldc < N || (residual && ldr < N)) return D3F_ERR_ARG;
    if (M == 0 || N == 0) return D3F_OK;
    if (!C || (K > 0 && (!A || !B))) return D3F_ERR_ARG;
    GemmEpi E{row_scale, col_scale, col_shift, residual, ldr, leaky, alpha};
    GemmGather G{nullptr, 0, M, nullptr, nullptr, 0, K};
    return gemm_run(A, lda, B, ldb, C, ldc, M, N, K, E, G, workspace, workspace_bytes, M_dev, M_hint, (hipStream_t)stream_);
}

// Decoder step of models/D3Feat.py:39-63 + the unary block that follows it (models/network_blocks.py:207-219), one launch:
//   out = act( ([ x'[idx[m,0]] | skip[m] ] @ W) * col_scale + col_shift ),   x' = x + zero row (closest_pool :69-83)
// x f32[N1,C1] (ldx), idx i32[M, ld_idx] (column 0 used), skip f32[M,C2] (lds; may be NULL with C2 = 0), W f32[C1+C2, N].
extern "C" int d3f_gemm_upsample_cat_f32(const float* x, int N1, int ldx, int C1, const int* idx, int ld_idx,
                                         const float* skip, int lds, int C2, const float* W, int ldb, float* C, int ldc,
                                         int M, int N, const float* col_scale, const float* col_shift, int leaky, float alpha,
                                         void* workspace, size_t workspace_bytes, const int* M_dev, const int* N1_dev,
                                         int M_hint, void* stream_) {
    if (M < 0 || N < 0 || N1 < 0 || C1 < 1 || C2 < 0 || ldx < C1 || (idx && ld_idx < 1) || ldb < N || ldc < N ||
        (C2 > 0 && lds < C2))
        return D3F_ERR_ARG;
    if (C2 > 0 && (C1 % 4 != 0)) return D3F_ERR_ARG;
    if (M == 0 || N == 0) return D3F_OK;
    if (!x || !W || !C || (C2 > 0 && !skip)) return D3F_ERR_ARG;   // idx == NULL: rows of x are used in place (N1 >= M)
    if (!idx && N1 < M) return D3F_ERR_ARG;
    GemmEpi E{nullptr, col_scale, col_shift, nullptr, 0, leaky, alpha};
    GemmGather G{idx, ld_idx, N1, N1_dev, C2 > 0 ? skip : nullptr, lds, C1};
    if (C2 > 0 && ((lds % 4 != 0) || (((uintptr_t)skip & 15) != 0))) return D3F_ERR_ARG;
    return gemm_run(x, ldx, W, ldb, C, ldc, M, N, C1 + C2, E, G, workspace, workspace_bytes, M_dev, M_hint, (hipStream_t)stream_);
}

static int gemm_run(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, GemmEpi E,
                    GemmGather G, void* workspace, size_t workspace_bytes, const int* M_dev, int M_hint, hipStream_t stream) {
    int bm, bn, S, tps;
    gemm_plan(M, N, K > 0 ? K : 1, M_hint, bm, bn, S, tps);
    float* slab = nullptr;
    if (S > 1) {
        const size_t need = (size_t)S * M * N * sizeof(float);
        if (!workspace || workspace_bytes < need) return D3F_ERR_WORKSPACE;
        slab = (float*)workspace;
    }
    const int vecA = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const int vecB = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    if (d3f_cdiv(M, bm) > 65535) return D3F_ERR_ARG;
    dim3 grid(d3f_cdiv(N, bn), S, d3f_cdiv(M, bm));
    // float4-addressable operands (every shape of the network): the straight-line kernel; anything else: the generic one
    const bool fast = (!E.residual || (E.ldr % 4 == 0 && ((uintptr_t)E.residual & 15) == 0)) && vecA && vecB &&
                      K % 4 == 0 && N % 4 == 0 && ldc % 4 == 0 &&
                      (((uintptr_t)C | (uintptr_t)E.col_scale | (uintptr_t)E.col_shift) & 15) == 0 &&
                      (!G.A2 || (G.K1 % 4 == 0 && G.lda2 % 4 == 0 && ((uintptr_t)G.A2 & 15) == 0));
    if (fast) {
        const size_t lds = (size_t)2 * (bm + bn) * GF_S * sizeof(float);
#define D3F_GEMM_E(WM_, WN_, TM_, TN_, NA_, EPI_)                                                                     \
    gemm_fast_kernel<WM_, WN_, TM_, TN_, NA_, EPI_><<<grid, 256, lds, stream>>>(A, lda, B, ldb, C, ldc, M, N, K, tps, slab, E, \
                                                                                M_dev, G)
#define D3F_GEMM(WM_, WN_, TM_, TN_, NA_)                                                                              \
    do {                                                                                                               \
        const int epi = (E.row_scale ? 1 : 0) | (E.residual ? 2 : 0);                                                  \
        if (epi == 0) D3F_GEMM_E(WM_, WN_, TM_, TN_, NA_, 0);                                                          \
        else if (epi == 1) D3F_GEMM_E(WM_, WN_, TM_, TN_, NA_, 1);                                                     \
        else if (epi == 2) D3F_GEMM_E(WM_, WN_, TM_, TN_, NA_, 2);                                                     \
        else D3F_GEMM_E(WM_, WN_, TM_, TN_, NA_, 3);                                                                   \
    } while (0)
        if (bn == 32) D3F_GEMM(4, 1, 1, 1, 1);
        else D3F_GEMM(2, 2, 1, 1, 1);
#undef D3F_GEMM
#undef D3F_GEMM_E
    } else {
        if (bn != 32) { bm = 64; bn = 64; grid = dim3(d3f_cdiv(N, bn), S, d3f_cdiv(M, bm)); }
#define D3F_GEMM(WM_, WN_, TM_, TN_)                                                                                   \
    gemm_f32_kernel<WM_, WN_, TM_, TN_><<<grid, 256, 0, stream>>>(A, lda, B, ldb, C, ldc, M, N, K, vecA, vecB, tps, slab, E, \
                                                                  M_dev, G)
        if (bn == 32) D3F_GEMM(4, 1, 1, 1);
        else D3F_GEMM(2, 2, 1, 1);
#undef D3F_GEMM
    }
    if (S > 1)
        gemm_splitk_reduce_kernel<<<d3f_cdiv((long long)M * N, 256), 256, 0, stream>>>(slab, S, M, N, C, ldc, E, M_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ---- LDS-DMA form (gemm_dma.h): weights pre-transposed once, k-tiles in a STAGES-deep LDS ring ------------------------
extern "C" int d3f_gemm_pack_f32t(const float* B, int ldb, int K, int N, float* Wt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 1 || N < 1 || ldb < N || !B || !Wt) return D3F_ERR_ARG;
    const int Kp = (K + GD_BK - 1) / GD_BK * GD_BK;
    gemm_pack_f32t_kernel<<<d3f_cdiv((long long)N * Kp, 256), 256, 0, stream>>>(B, ldb, K, N, Kp, Wt);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// The measured variants of this launcher (2 / 3 / 4 ring stages, 128 x 64 and 128 x 128 register tiles, a persistent grid whose DMA
// ring runs across tile boundaries, a loader wavefront) are recorded in profiles/r04_experiments.txt g1-g6 with their numbers; the
// persistent walk is still in the kernel (grid = items is its degenerate case), the others live in git history /
// tools/ubench/gemm_dma_loader_wave.patch.  What ships: 3 stages, 64 x 64 tiles (128 x 32 when N <= 32), one item per workgroup.
#define GD_STAGES 3

// Same operator and argument meaning as d3f_gemm_bf16 (the union of d3f_gemm_f32 and d3f_gemm_upsample_cat_f32), in fp32:
//   C = act( ([ A'[idx[m,0]] | skip[m] ] @ W) * row_scale * col_scale + col_shift + residual ),  Wt = d3f_gemm_pack_f32t(W).
// Operands must be float4-addressable (C1, C2, lda, lds, ldc, ldr multiples of 4; 16-byte aligned bases) -- D3F_ERR_ARG otherwise:
// the caller then uses d3f_gemm_f32 / d3f_gemm_upsample_cat_f32, which take any shape.
extern "C" int d3f_gemm_f32t(const float* A, int N1, int lda, int C1, const int* idx, int ld_idx, const float* skip, int lds, int C2,
                             const float* Wt, float* C, int ldc, int M, int N, const float* row_scale, const float* col_scale,
                             const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* workspace,
                             size_t workspace_bytes, const int* M_dev, const int* N1_dev, int M_hint, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int K = C1 + C2;
    if (M < 0 || N < 1 || N1 < 0 || C1 < 4 || C2 < 0 || (C1 % 4) || (C2 % 4) || (N % 4) || lda < C1 || (lda % 4) || ldc < N || (ldc % 4) ||
        (C2 > 0 && (lds < C2 || (lds % 4))) || (residual && (ldr < N || (ldr % 4))) || (idx && ld_idx < 1) || (!idx && N1 < M))
        return D3F_ERR_ARG;
    if (M == 0) return D3F_OK;
    if (!A || !Wt || !C || (C2 > 0 && !skip) ||
        (((uintptr_t)A | (uintptr_t)Wt | (uintptr_t)C | (uintptr_t)skip | (uintptr_t)residual | (uintptr_t)col_scale | (uintptr_t)col_shift) & 15))
        return D3F_ERR_ARG;
    const int Kp = (K + GD_BK - 1) / GD_BK * GD_BK;
    int bm, bn, S, tps;
    gemm_plan(M, N, K, M_hint, bm, bn, S, tps);
    float* slab = nullptr;
    if (S > 1) {
        if (!workspace || workspace_bytes < (size_t)S * M * N * sizeof(float)) return D3F_ERR_WORKSPACE;
        slab = (float*)workspace;
    }
    GemmEpi E{row_scale, col_scale, col_shift, residual, ldr, leaky, alpha};
    GemmGather G{idx, ld_idx, N1, N1_dev, C2 > 0 ? skip : nullptr, lds, C1};
    const size_t lds_bytes = (size_t)GD_STAGES * (bm + bn) * GD_BK * sizeof(float);
    const long long items_cap = (long long)d3f_cdiv(N, bn) * S * d3f_cdiv(M, bm);
    const long long gsz = (items_cap + 7) / 8 * 8;        // one item per workgroup; a multiple of 8 keeps the XCD item order whole
    if (gsz > 0x7fffffffll) return D3F_ERR_ARG;
    dim3 grid((unsigned)gsz, 1, 1);
#define D3F_GD_E(WM_, WN_, TM_, TN_, ST_, EPI_)                                                                                 \
    do {                                                                                                                       \
        if (lds_bytes > 65536) {                                                                                               \
            static std::atomic<int> done{0};                                                                                   \
            if (!done.load(std::memory_order_acquire)) {                                                                       \
                if (hipFuncSetAttribute((const void*)gemm_dma_kernel<WM_, WN_, TM_, TN_, ST_, EPI_>,                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return D3F_ERR_HIP; \
                done.store(1, std::memory_order_release);                                                                      \
            }                                                                                                                  \
        }                                                                                                                      \
        gemm_dma_kernel<WM_, WN_, TM_, TN_, ST_, EPI_><<<grid, 256, lds_bytes, stream>>>(A, lda, Wt, Kp, C, ldc, M, N, K, tps, S, slab, \
                                                                                          E, M_dev, G);                        \
    } while (0)
#define D3F_GD_S(WM_, WN_, TM_, TN_, ST_)                                                                                       \
    do {                                                                                                                       \
        const int epi = (E.row_scale ? 1 : 0) | (E.residual ? 2 : 0);                                                          \
        if (epi == 0) D3F_GD_E(WM_, WN_, TM_, TN_, ST_, 0);                                                                    \
        else if (epi == 1) D3F_GD_E(WM_, WN_, TM_, TN_, ST_, 1);                                                               \
        else if (epi == 2) D3F_GD_E(WM_, WN_, TM_, TN_, ST_, 2);                                                               \
        else D3F_GD_E(WM_, WN_, TM_, TN_, ST_, 3);                                                                             \
    } while (0)
    if (bn == 32) D3F_GD_S(4, 1, 1, 1, GD_STAGES);
    else D3F_GD_S(2, 2, 1, 1, GD_STAGES);
#undef D3F_GD
#undef D3F_GD_S
#undef D3F_GD_E
    if (S > 1)
        gemm_splitk_reduce_kernel<<<d3f_cdiv((long long)M * N, 256), 256, 0, stream>>>(slab, S, M, N, C, ldc, E, M_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// =====================================================================================================================
// bf16-operand contraction (BASELINE.json configs[4]: batched inference, "bf16 features with MFMA contraction").
//
// Same operator as d3f_gemm_f32 / d3f_gemm_upsample_cat_f32 -- same composite A operand, same epilogue -- but both operands are
// rounded to bfloat16 (round to nearest even) on their way into LDS and multiplied with v_mfma_f32_32x32x16_bf16, fp32
// accumulate: 16x the fp32 matrix rate, which turns every contraction of the network from matrix-pipe bound into memory
// bound.  NOT the fp32 path's arithmetic: each product carries the 2^-9 relative rounding of its operands, so results are
// compared with the fp32 oracle at a documented, looser tolerance (tests/test_gpu_bf16.py) and are reported as a separate
// bench configuration, never as the fp32 headline.
//   A  f32 (activations stay fp32 in HBM; converted while staging)     W  pre-packed once: bf16 [N][Kp], k contiguous,
//   Kp = K rounded up to 32 and zero padded (d3f_gemm_pack_bf16), so A and W tiles are staged by the same code:
//   rows = output index, 32 k-values = 64 bytes per row and stage, row stride 80 bytes in LDS (16-byte aligned fragments,
//   the 16 lanes of a ds_read_b128 group on 16 distinct 4-bank slots).
// Workgroup 256 threads = 2 x 2 wavefronts x one 32x32 accumulator tile; BK = 32 (two MFMAs per stage and wave), double
// buffered.  Skinny deep layers reuse the fp32 kernel's K split (slabs + ordered reduction).
// =====================================================================================================================
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
#define GB_BK 32
#define GB_LS 40          // LDS row stride in bf16 elements (80 bytes)

__device__ __forceinline__ unsigned gb_rne(float f) {          // fp32 -> bf16 bits, round to nearest even (NaN stays NaN)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned gb_pack2(float a, float b) { return gb_rne(a) | (gb_rne(b) << 16); }

__global__ void __launch_bounds__(256) gemm_pack_bf16_kernel(const float* __restrict__ B, int ldb, int K, int N, int Kp,
                                                             unsigned short* __restrict__ Wt) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * Kp) return;
    const int n = (int)(t / Kp), k = (int)(t % Kp);
    Wt[t] = (unsigned short)(k < K ? gb_rne(B[(size_t)k * ldb + n]) : 0u);
}

extern "C" int d3f_gemm_pack_bf16(const float* B, int ldb, int K, int N, void* Wt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 1 || N < 1 || ldb < N || !B || !Wt) return D3F_ERR_ARG;
    const int Kp = (K + GB_BK - 1) / GB_BK * GB_BK;
    gemm_pack_bf16_kernel<<<d3f_cdiv((long long)N * Kp, 256), 256, 0, stream>>>(B, ldb, K, N, Kp, (unsigned short*)Wt);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

template <bool ABF, bool CBF>   // ABF: A / skip / residual hold bfloat16 (bf16 feature storage); CBF: C is written as bfloat16
__global__ void __launch_bounds__(256)
gemm_bf16_kernel(const float* __restrict__ A, int lda, const unsigned short* __restrict__ Wt, int Kp, float* __restrict__ C,
                 int ldc, int M, int N, int K, int tiles_per_split, float* __restrict__ slab, GemmEpi E,
                 const int* __restrict__ M_dev, GemmGather G) {
    constexpr int BM = 64, BN = 64;
    const int Mcap = M;
    M = d3f_dyn(M, M_dev);
    if ((int)(blockIdx.z * BM) >= M) return;   // capacity-sized grid (row tile = slowest dispatch dimension, as gemm_fast_kernel)
    // live workgroups = the first contiguous run of the dispatch order (row tile is the slowest grid dimension); among them
    // every XCD takes one contiguous run of (row tile, K slice, column tile) triples, so the column tiles of a row block --
    // which read the same A rows -- meet in one L2 (common.h: d3f_xcd_tile)
    const unsigned gx_ = gridDim.x, gxy_ = gridDim.x * gridDim.y;
    const unsigned T_ = d3f_xcd_tile(blockIdx.x + gx_ * blockIdx.y + gxy_ * blockIdx.z, gxy_ * (unsigned)((M + BM - 1) / BM));
    const unsigned bz = T_ / gxy_, by = (T_ % gxy_) / gx_, bx = T_ % gx_;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * GB_LS];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * GB_LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = bz * BM, n0 = bx * BN;
    const int nt_all = Kp / GB_BK;
    const int t_begin = by * tiles_per_split;
    const int t_end = min(nt_all, t_begin + tiles_per_split);
    // staging roles: A tile 64 rows x 32 k fp32 = 512 float4 slots -> 2 per thread (row = slot / 8, k4 = slot % 8);
    //                W tile 64 rows x 32 k bf16 = 256 uint4 slots (8 bf16 each) -> 1 per thread (row = tid / 4, k8 = tid % 4)
    int srow[2];
    {
        const int n1 = d3f_dyn(G.N1, G.N1_dev);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = m0 + ((tid + i * 256) >> 3);
            int sr = gm < M ? gm : -1;
            if (G.gidx && sr >= 0) {
                sr = G.gidx[(size_t)gm * G.ld_gidx];
                if (sr < 0 || sr >= n1) sr = -1;      // shadow neighbour: zero row
            }
            srow[i] = sr;
        }
    }
    const int brow = n0 + (tid >> 2);
    float4 ra[2];      // fp32 activations: four values per slot, rounded while staging
    uint2 rh[2];       // bf16 activations: the same four values, already in storage format
    uint4 rb;
    const unsigned short* Ah = (const unsigned short*)A;
    const unsigned short* A2h = (const unsigned short*)G.A2;
    auto load_tile = [&](int t) {
        const int k0 = t * GB_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            const int gm = m0 + (e >> 3), k = k0 + ((e & 7) << 2);
            if (ABF) {
                uint2 v = make_uint2(0u, 0u);
                if (gm < M && k < K) {
                    if (G.A2 && k >= G.K1) v = *(const uint2*)(A2h + (size_t)gm * G.lda2 + (k - G.K1));
                    else if (srow[i] >= 0) v = *(const uint2*)(Ah + (size_t)srow[i] * lda + k);
                }
                rh[i] = v;
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gm < M && k < K) {             // (K, K1 and the leading dimensions are multiples of 4: checked by the launcher)
                    if (G.A2 && k >= G.K1) v = *(const float4*)(G.A2 + (size_t)gm * G.lda2 + (k - G.K1));
                    else if (srow[i] >= 0) v = *(const float4*)(A + (size_t)srow[i] * lda + k);
                }
                ra[i] = v;
            }
        }
        rb = (brow < N) ? *(const uint4*)(Wt + (size_t)brow * Kp + k0 + ((tid & 3) << 3)) : make_uint4(0u, 0u, 0u, 0u);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            const uint2 p = ABF ? rh[i] : make_uint2(gb_pack2(ra[i].x, ra[i].y), gb_pack2(ra[i].z, ra[i].w));
            *(uint2*)&As[buf][(e >> 3) * GB_LS + ((e & 7) << 2)] = p;
        }
        *(uint4*)&Bs[buf][(tid >> 2) * GB_LS + ((tid & 3) << 3)] = rb;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const int buf = (t - t_begin) & 1;
            if (t + 1 < t_end) load_tile(t + 1);
            // fragments: lane (r = lane & 31, h = lane >> 5) holds k = 8h .. 8h+7 of a 16-deep MFMA step
            const unsigned short* ap = &As[buf][(32 * wm + (lane & 31)) * GB_LS + ((lane >> 5) << 3)];
            const unsigned short* bp = &Bs[buf][(32 * wn + (lane & 31)) * GB_LS + ((lane >> 5) << 3)];
            const uint4 a0 = *(const uint4*)ap, a1 = *(const uint4*)(ap + 16);
            const uint4 b0 = *(const uint4*)bp, b1 = *(const uint4*)(bp + 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, a0), __builtin_bit_cast(gb_bf16x8, b0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, a1), __builtin_bit_cast(gb_bf16x8, b1), acc, 0, 0, 0);
            if (t + 1 < t_end) store_tile(buf ^ 1);
            __syncthreads();
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int gn = n0 + 32 * wn + (lane & 31);
    if (gn >= N) return;
    float cs = 1.f, ch = 0.f;
    if (!slab) { cs = E.col_scale ? E.col_scale[gn] : 1.f; ch = E.col_shift ? E.col_shift[gn] : 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm >= M) continue;
        if (slab) { slab[((size_t)by * Mcap + gm) * N + gn] = acc[r]; continue; }
        float v = acc[r];
        if (E.row_scale) v *= E.row_scale[gm];
        v = v * cs + ch;
        if (E.residual)
            v += ABF ? d3f_bf16_f32(((const unsigned short*)E.residual)[(size_t)gm * E.ldr + gn]) : E.residual[(size_t)gm * E.ldr + gn];
        if (E.leaky) v = v > 0.f ? v : v * E.alpha;
        if (CBF) ((unsigned short*)C)[(size_t)gm * ldc + gn] = (unsigned short)d3f_bf16_rne(v);
        else C[(size_t)gm * ldc + gn] = v;
    }
}

// K split of the bf16 contraction (its tile is 64 x 64 whatever N is; the fp32 plan only decides the number of K slices)
static void gemm_bf16_split(int M, int N, int K, int M_hint, int& S, int& tps) {
    int bm, bn;
    gemm_plan(M, N > 32 ? N : 64, K, M_hint, bm, bn, S, tps);
    const int nt = ((K + GB_BK - 1) / GB_BK * GB_BK) / GB_BK;
    if (S > nt) S = nt;
    tps = d3f_cdiv(nt, S);
    S = d3f_cdiv(nt, tps);
}

extern "C" size_t d3f_gemm_bf16_workspace_bytes(int M, int N, int K, int M_hint) {
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    int S, tps;
    gemm_bf16_split(M, N, K, M_hint, S, tps);
    return S > 1 ? d3f_align((size_t)S * M * N * sizeof(float)) + 256 : 256;
}

// A f32[M,K] (lda) or the composite [ x'[idx[m,0]] | skip[m] ] (idx != NULL / skip != NULL, as d3f_gemm_upsample_cat_f32);
// Wt = d3f_gemm_pack_bf16(W [K,N]).  K, K1 = C1, lda, lds multiples of 4 and 16-byte aligned bases (else D3F_ERR_ARG: use the
// fp32 entry points).  workspace >= d3f_gemm_bf16_workspace_bytes(M, N, K, M_hint).
extern "C" int d3f_gemm_bf16(const void* A_, int N1, int lda, int C1, const int* idx, int ld_idx, const void* skip_, int lds,
                             int C2, const void* Wt, void* C_, int ldc, int M, int N, const float* row_scale,
                             const float* col_scale, const float* col_shift, const void* residual_, int ldr, int leaky,
                             float alpha, void* workspace, size_t workspace_bytes, const int* M_dev, const int* N1_dev,
                             int M_hint, int a_bf16, int c_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const float* A = (const float*)A_;
    const float* skip = (const float*)skip_;
    const float* residual = (const float*)residual_;
    float* C = (float*)C_;
    const int K = C1 + C2;
    if (M < 0 || N < 1 || N1 < 0 || C1 < 4 || C2 < 0 || (K % 4) || (C1 % 4) || lda < C1 || (lda % 4) || ldc < N ||
        (C2 > 0 && (lds < C2 || (lds % 4))) || (residual && ldr < N) || (idx && ld_idx < 1) || (!idx && N1 < M))
        return D3F_ERR_ARG;
    if (M == 0) return D3F_OK;
    if (!A || !Wt || !C || (C2 > 0 && !skip) || ((uintptr_t)Wt & 15) || (((uintptr_t)A | (uintptr_t)skip) & (a_bf16 ? 7 : 15)))
        return D3F_ERR_ARG;
    const int Kp = (K + GB_BK - 1) / GB_BK * GB_BK;
    int S, tps;
    gemm_bf16_split(M, N, K, M_hint, S, tps);
    float* slab = nullptr;
    if (S > 1) {
        if (!workspace || workspace_bytes < (size_t)S * M * N * sizeof(float)) return D3F_ERR_WORKSPACE;
        slab = (float*)workspace;
    }
    if (d3f_cdiv(M, 64) > 65535) return D3F_ERR_ARG;
    GemmEpi E{row_scale, col_scale, col_shift, residual, ldr, leaky, alpha};
    GemmGather G{idx, ld_idx, N1, N1_dev, C2 > 0 ? skip : nullptr, lds, C1};
    dim3 grid(d3f_cdiv(N, 64), S, d3f_cdiv(M, 64));
#define D3F_GB(ABF_, CBF_) gemm_bf16_kernel<ABF_, CBF_><<<grid, 256, 0, stream>>>(A, lda, (const unsigned short*)Wt, Kp, C, ldc, M, N, K, tps, slab, E, M_dev, G)
    if (a_bf16) { if (c_bf16) D3F_GB(true, true); else D3F_GB(true, false); }
    else { if (c_bf16) D3F_GB(false, true); else D3F_GB(false, false); }
#undef D3F_GB
    if (S > 1)
        gemm_splitk_reduce_kernel<<<d3f_cdiv((long long)M * N, 256), 256, 0, stream>>>(slab, S, M, N, C, ldc, E, M_dev, a_bf16, c_bf16);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// =====================================================================================================================
// fp32 contraction by exact operand splitting on the bf16 matrix cores (gemm_x3.h)
// =====================================================================================================================
#include "gemm_x3.h"

// Tile and K split.  Three workgroup shapes: 128 rows x 32 / 64 columns (4 wavefronts, 2 resident per CU) and 256 x 128 (8
// wavefronts, 1 per CU: a quarter of the operand traffic per product -- both forms are bound by what the CUs can pull out of L2,
// ~9 TB/s -- but a quarter of the workgroups).  For each candidate the K slice count that minimises
//   rounds(workgroups / resident slots) x (k-tiles per slice + 12: a workgroup's fixed cost) x its time per k-tile  +  2 per slab,
// in units of one k-tile of the 128 x 64 form (the 256 x 128 form's k-tile costs 1.63 of them: 3.1 against 1.9 us at M = 65536,
// K = 3072, N = 256); the cheaper candidate wins.  E.g. 288 workgroups x 96 k-tiles (M = 4525, K = 3072, N = 512) are ONE round
// of 96 unsplit, two rounds of 48 halved (576 > 512 slots), 2 x 32 in thirds -- and one round of 32 x 1.63 as 72 x 3 big ones.
// (The constants are end-to-end fits, profiles/r04_experiments.txt x10: fixed cost 4 .. 16 k-tiles, 256 x 128 factor 1.0 .. 1.63 and
// 2 .. 4 per slab are one plateau; 2.0 for the factor or 1 per slab lose 1-3 %.)
struct GemmX3Plan { int tn, waves, S, tps; };
static long long gemm_x3_cost(long long blocks, long long slots, int nt, int per_tile_x100, int& S) {
    long long best = -1;
    S = 1;
    for (int s = 1; s <= 64 && s * 4 <= nt; ++s) {
        const int t = d3f_cdiv(nt, s);
        if (d3f_cdiv(nt, t) != s) continue;
        const long long rounds = (blocks * s + slots - 1) / slots;
        const long long cost = rounds * (t + 12) * per_tile_x100 + (s > 1 ? 200ll * s : 0);
        if (best < 0 || cost < best) { best = cost; S = s; }
    }
    if (best < 0) best = ((blocks + slots - 1) / slots) * (nt + 12) * per_tile_x100;     // fewer than 4 k-tiles: never split
    return best;
}
static GemmX3Plan gemm_x3_plan(int M, int N, int K, int M_hint) {
    if (M_hint > 0 && M_hint < M) M = M_hint;
    const int nt = K / GX_BK;
    GemmX3Plan p;
#ifdef D3F_GEMM_TUNING      // measurement builds only (tools/ubench/build_variant.sh -DD3F_GEMM_TUNING): never in the shipped library
    // D3F_X3_PLAN="tn,waves,S" forces the workgroup shape (1,4 / 2,4 / 4,8) and the K slice count of every call it is valid for
    static const char* forced = getenv("D3F_X3_PLAN");
    if (forced) {
        int tn = 0, wv = 0, S = 0;
        if (sscanf(forced, "%d,%d,%d", &tn, &wv, &S) == 3 && ((tn == 1 && wv == 4) || (tn == 2 && wv == 4) || (tn == 4 && wv == 8)) && S >= 1 &&
            !(N <= 32 && tn != 1) && S * 4 <= (nt > 4 ? nt : 4)) {       // (the shapes the cost model itself may choose)
            p.tn = tn; p.waves = wv; p.S = S < nt ? S : nt;
            p.tps = d3f_cdiv(nt, p.S);
            p.S = d3f_cdiv(nt, p.tps);
            return p;
        }
    }
#endif
    p.tn = N <= 32 ? 1 : 2;
    p.waves = 4;
    const long long c_small = gemm_x3_cost((long long)d3f_cdiv(M, 128) * d3f_cdiv(N, 32 * p.tn), 512, nt, 100, p.S);
    if (N >= 128) {
        int s_big;
        const long long c_big = gemm_x3_cost((long long)d3f_cdiv(M, 256) * d3f_cdiv(N, 128), 256, nt, 163, s_big);
        if (c_big < c_small) { p.tn = 4; p.waves = 8; p.S = s_big; }
    }
    p.tps = d3f_cdiv(nt, p.S);
    p.S = d3f_cdiv(nt, p.tps);
    return p;
}

extern "C" size_t d3f_gemm_x3_packed_bytes(int K, int N) {
    if (K < 1 || N < 1) return 0;
    return (size_t)d3f_cdiv(N, 32) * d3f_cdiv(K, GX_BK) * GX_CHUNK * sizeof(unsigned short);
}

extern "C" int d3f_gemm_pack_x3(const float* B, int ldb, int K, int N, void* Wx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 1 || N < 1 || ldb < N || !B || !Wx || ((uintptr_t)Wx & 15)) return D3F_ERR_ARG;
    const int nkt = d3f_cdiv(K, GX_BK);
    const long long total = (long long)d3f_cdiv(N, 32) * nkt * GX_CHUNK;
    gemm_pack_x3_kernel<<<d3f_cdiv(total, 256), 256, 0, stream>>>(B, ldb, K, N, nkt, total, (unsigned short*)Wx);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// the plan d3f_gemm_x3 will use for a problem: rows / columns of its workgroups and the K slice count (diagnostics and tests)
extern "C" int d3f_gemm_x3_plan(int M, int N, int K, int M_hint, int* rows, int* cols, int* slices) {
    if (M <= 0 || N <= 0 || K < GX_BK || (K % GX_BK)) return D3F_ERR_ARG;
    const GemmX3Plan p = gemm_x3_plan(M, N, K, M_hint);
    if (rows) *rows = 32 * p.waves;
    if (cols) *cols = 32 * p.tn;
    if (slices) *slices = p.S;
    return D3F_OK;
}

extern "C" size_t d3f_gemm_x3_workspace_bytes(int M, int N, int K, int M_hint) {
    if (M <= 0 || N <= 0 || K < GX_BK) return 256;
    const GemmX3Plan p = gemm_x3_plan(M, N, K, M_hint);
    return p.S > 1 ? d3f_align((size_t)p.S * M * N * sizeof(float)) + 256 : 256;
}

// Same operator and argument meaning as d3f_gemm_f32t; Wx = d3f_gemm_pack_x3(W [K,N]).  On top of d3f_gemm_f32t's addressing rules:
// K = C1 + C2 a multiple of 32 and, for a concatenated operand, C1 a multiple of 32 too (else D3F_ERR_ARG: use d3f_gemm_f32t).
// workspace >= d3f_gemm_x3_workspace_bytes(M, N, K, M_hint).
extern "C" int d3f_gemm_x3(const float* A, int N1, int lda, int C1, const int* idx, int ld_idx, const float* skip, int lds, int C2,
                           const void* Wx, float* C, int ldc, int M, int N, const float* row_scale, const float* col_scale,
                           const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* workspace,
                           size_t workspace_bytes, const int* M_dev, const int* N1_dev, int M_hint, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int K = C1 + C2;
    if (M < 0 || N < 1 || N1 < 0 || C1 < 4 || C2 < 0 || (C1 % 4) || (C2 % 4) || (N % 4) || lda < C1 || (lda % 4) || ldc < N || (ldc % 4) ||
        (C2 > 0 && (lds < C2 || (lds % 4))) || (residual && (ldr < N || (ldr % 4))) || (idx && ld_idx < 1) || (!idx && N1 < M))
        return D3F_ERR_ARG;
    if ((K % GX_BK) || (C2 > 0 && (C1 % GX_BK))) return D3F_ERR_ARG;
    if (M == 0) return D3F_OK;
    if (!A || !Wx || !C || (C2 > 0 && !skip) ||
        (((uintptr_t)A | (uintptr_t)Wx | (uintptr_t)C | (uintptr_t)skip | (uintptr_t)residual | (uintptr_t)col_scale | (uintptr_t)col_shift) & 15))
        return D3F_ERR_ARG;
    GemmEpi E{row_scale, col_scale, col_shift, residual, ldr, leaky, alpha};
    GemmGather G{idx, ld_idx, N1, N1_dev, C2 > 0 ? skip : nullptr, lds, C1};
    const int nkt = K / GX_BK, NG = d3f_cdiv(N, 32);
    {   // ---- the resident-W persistent form: tall contractions whose whole pre-split W fits the LDS of a CU (gemm_x3.h) ----
        static const bool on = []() { const char* e = getenv("D3F_GEMM_X3R"); return !(e && e[0] == '0'); }();
        const int m_eff = (M_hint > 0 && M_hint < M) ? M_hint : M;
        const size_t wbytes = (size_t)nkt * NG * GX_CHUNK * sizeof(unsigned short);
        if (on && (NG == 1 || NG == 2 || NG == 4) && wbytes + GXR_PATCH_BYTES + GXR_EPI_BYTES <= 160 * 1024 && m_eff >= 65536) {
            // CU count of the CURRENT device (cached per device ordinal: a process may drive several)
            static std::atomic<int> cus_of[64];
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) return D3F_ERR_HIP;
            int cus = cus_of[dev & 63].load(std::memory_order_relaxed);
            if (!cus) {
                hipDeviceProp_t pr;
                if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return D3F_ERR_HIP;
                cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
                cus_of[dev & 63].store(cus, std::memory_order_relaxed);
            }
            static std::atomic<unsigned long long> lds_done{0};
            const void* const fns[3] = {(const void*)gemm_x3r_kernel<1>, (const void*)gemm_x3r_kernel<2>, (const void*)gemm_x3r_kernel<4>};
            if (d3f_opt_in_lds(lds_done, fns, 160 * 1024) != D3F_OK) return D3F_ERR_HIP;
            const int grid = std::min(cus, d3f_cdiv(d3f_cdiv(M, 32), 8));
#define D3F_GXR(TN_) gemm_x3r_kernel<TN_><<<grid, 512, wbytes + GXR_PATCH_BYTES + GXR_EPI_BYTES, stream>>>(A, lda, (const unsigned short*)Wx, nkt, C, ldc, M, N, E, M_dev, G)
            if (NG == 4) D3F_GXR(4);
            else if (NG == 2) D3F_GXR(2);
            else D3F_GXR(1);
#undef D3F_GXR
            D3F_LAUNCH_CHECK();
            return D3F_OK;
        }
    }
    const GemmX3Plan pl = gemm_x3_plan(M, N, K, M_hint);
    const int S = pl.S, tps = pl.tps, bm = 32 * pl.waves;
    float* slab = nullptr;
    if (S > 1) {
        if (!workspace || workspace_bytes < (size_t)S * M * N * sizeof(float)) return D3F_ERR_WORKSPACE;
        slab = (float*)workspace;
    }
    if (d3f_cdiv(M, bm) > 65535) return D3F_ERR_ARG;
    dim3 grid(d3f_cdiv(N, 32 * pl.tn), S, d3f_cdiv(M, bm));
#define D3F_GX(TN_, WV_) gemm_x3_kernel<TN_, WV_><<<grid, 64 * WV_, 0, stream>>>(A, lda, (const unsigned short*)Wx, nkt, NG, C, ldc, M, N, tps, slab, E, M_dev, G)
    if (pl.waves == 8) D3F_GX(4, 8);
    else if (pl.tn == 1) D3F_GX(1, 4);
    else D3F_GX(2, 4);
#undef D3F_GX
    if (S > 1)
        gemm_splitk_reduce_kernel<<<d3f_cdiv((long long)M * N, 256), 256, 0, stream>>>(slab, S, M, N, C, ldc, E, M_dev);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}
