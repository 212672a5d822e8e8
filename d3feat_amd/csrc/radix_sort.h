// Stable LSD radix sort of (u32 key, u32 payload) pairs for gfx950, hand written for the stage-0 voxel pass
// (grid_subsample.hip): 8-bit digits, two launches per digit pass, every size read from HBM.
//
//   * The number of items n and the number of significant key bits live on the DEVICE (`RsMeta`): a captured launch
//     sequence always issues RS_MAX_PASSES passes; a pass whose digit lies above the significant bits returns at once,
//     so a 21-bit key (3DMatch room: 18 voxel-key bits + 3 element bits) costs 3 real passes, a KITTI sweep 4.
//   * Pass p reads buffer (p & 1) and writes buffer ((p + 1) & 1); the sorted data end up in buffer (npass & 1).
//     Pass 0 takes the payload to be the item's position (no index array is read).
//   * rs_hist_kernel: one 8192-item tile per workgroup, digit histogram in LDS -> hist[tile][256] (tile-major rows of 1 KB).
//     The histogram of pass 0 is produced by the caller's key kernel (`rs_tile_histogram`), which has the keys in
//     registers anyway.
//   * rs_scatter_kernel: every workgroup sums the columns of the (tiles x 256) matrix itself -- everything before its own
//     tile and the column totals; wave w takes rows w, w + 8, ..., one 16-byte load per lane = one 1 KB row per instruction,
//     8 rows in flight (a thread-per-column loop was 73 dependent L2 round trips: 168 us for 1.2 M items) -- which replaces a
//     separate scan launch and any cross-workgroup hand-off; digit bases by one 256-wide scan.  Ranking is wave-local and stable: a wave owns 1024
//     consecutive items and walks them in 16 rounds of 64; the lanes of a round that hold the same digit find each other
//     with 8 wavefront ballots (one per digit bit), rank = popcount of the peers below, and the per-wave digit counter in
//     LDS advances by the peer count (the wave's LDS accesses are in program order: all peers read the counter, then the
//     lowest peer writes it).  Order inside a digit = (tile, wave, round, lane) = input order.  The tile is then put in
//     digit order in LDS (64 KB) and written out position by position: one contiguous run per digit and tile.
//   * No resets between passes or replays: every word of `hist` that a pass reads was written by the same pass.
#pragma once
#include "common.h"

#define RS_THREADS 512
#define RS_WAVES (RS_THREADS / 64)
#define RS_ROUNDS 16
#define RS_TILE (RS_THREADS * RS_ROUNDS)   // 8192 items per workgroup
#define RS_WAVE_ITEMS (64 * RS_ROUNDS)     // 1024 consecutive items per wave
#define RS_MAX_PASSES 4

struct RsMeta {       // device-resident description of one sort (written by the caller's prologue kernel)
    int n;            // items
    int bits;         // significant key bits (1..32)
    int npass;        // ceil(bits / 8)
    int kb;           // caller's field: bit position of the element field inside the key
};

#define RS_SCATTER_LDS_BYTES ((2 * RS_WAVES * 256 + 256 + 16 + 2 * RS_TILE) * 4)
static inline int rs_tiles(int n_cap) { return d3f_cdiv(n_cap > 0 ? n_cap : 1, RS_TILE); }
static inline size_t rs_hist_words(int n_cap) { return (size_t)rs_tiles(n_cap) * 256; }

#ifdef __HIPCC__
// digit histogram of one tile from registers: key[r] is the item at tile position w * 1024 + r * 64 + lane (valid[r] = it
// exists).  sHist: 256 words of LDS.  All RS_THREADS threads of the workgroup call this.
__device__ __forceinline__ void rs_tile_histogram(const unsigned (&key)[RS_ROUNDS], unsigned valid_mask, int shift,
                                                  unsigned* __restrict__ sHist, unsigned* __restrict__ hist_row) {
    if (threadIdx.x < 256) sHist[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r)
        if (valid_mask & (1u << r)) atomicAdd(&sHist[(key[r] >> shift) & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < 256) hist_row[threadIdx.x] = sHist[threadIdx.x];
}

__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const RsMeta* __restrict__ meta, int pass,
                                                             const unsigned* __restrict__ key0,
                                                             const unsigned* __restrict__ key1,
                                                             unsigned* __restrict__ hist) {
    __shared__ unsigned sHist[256];
    const int n = meta->n;
    if (pass >= meta->npass) return;
    const int tile = blockIdx.x;
    if ((long long)tile * RS_TILE >= (long long)n) return;
    const unsigned* __restrict__ src = (pass & 1) ? key1 : key0;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = tile * RS_TILE + w * RS_WAVE_ITEMS + lane;
    unsigned key[RS_ROUNDS], vm = 0u;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int pos = base + r * 64;
        key[r] = 0u;
        if (pos < n) { key[r] = src[pos]; vm |= 1u << r; }
    }
    rs_tile_histogram(key, vm, pass * 8, sHist, hist + (size_t)tile * 256);
}

__global__ void __launch_bounds__(RS_THREADS) rs_scatter_kernel(const RsMeta* __restrict__ meta, int pass,
                                                                unsigned* __restrict__ key0, unsigned* __restrict__ key1,
                                                                unsigned* __restrict__ val0, unsigned* __restrict__ val1,
                                                                const unsigned* __restrict__ hist) {
    // 82 KB of LDS (dynamic: above the 64 KB static limit), carved by hand
    extern __shared__ __attribute__((aligned(16))) unsigned rs_lds[];
    unsigned (*sCnt)[256] = (unsigned (*)[256])rs_lds;                        // per wave: running digit counters, then the wave's base
    unsigned (*sPart)[256] = (unsigned (*)[256])(rs_lds + RS_WAVES * 256);    // per wave: partial column sums of the tiles before this one
    unsigned* sBase = rs_lds + 2 * RS_WAVES * 256;   // global position of this tile's first item of each digit, then the delta
    unsigned* sScan = sBase + 256;                   // RS_WAVES words (+ padding to 16)
    unsigned* sKey = sScan + 16;                     // the tile ordered by digit
    unsigned* sVal = sKey + RS_TILE;
    const int n = meta->n;
    if (pass >= meta->npass) return;
    const int tile = blockIdx.x;
    if ((long long)tile * RS_TILE >= (long long)n) return;
    const int live = (n + RS_TILE - 1) / RS_TILE;
    const unsigned* __restrict__ ksrc = (pass & 1) ? key1 : key0;
    const unsigned* __restrict__ vsrc = (pass & 1) ? val1 : val0;
    unsigned* __restrict__ kdst = (pass & 1) ? key0 : key1;
    unsigned* __restrict__ vdst = (pass & 1) ? val0 : val1;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int shift = pass * 8;

    // the items first: their loads are in flight while the histogram columns are summed
    const int base = tile * RS_TILE + w * RS_WAVE_ITEMS + lane;
    unsigned key[RS_ROUNDS], val[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int pos = base + r * 64;
        key[r] = 0u;
        val[r] = (unsigned)pos;
        if (pos < n) {
            key[r] = ksrc[pos];
            if (pass > 0) val[r] = vsrc[pos];
        }
    }
    // columns of the histogram matrix: items of every digit in the tiles before this one, and in all tiles
    {
        uint4 bef = make_uint4(0u, 0u, 0u, 0u), tot = make_uint4(0u, 0u, 0u, 0u);
        const uint4* __restrict__ rows = (const uint4*)hist + lane;      // lane l: digits 4 l .. 4 l + 3 of a row
        int t = w;
        for (; t + 7 * RS_WAVES < live; t += 8 * RS_WAVES) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = rows[(size_t)(t + u * RS_WAVES) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tot.x += v[u].x; tot.y += v[u].y; tot.z += v[u].z; tot.w += v[u].w;
                if (t + u * RS_WAVES < tile) { bef.x += v[u].x; bef.y += v[u].y; bef.z += v[u].z; bef.w += v[u].w; }
            }
        }
        for (; t < live; t += RS_WAVES) {
            const uint4 v = rows[(size_t)t * 64];
            tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            if (t < tile) { bef.x += v.x; bef.y += v.y; bef.z += v.z; bef.w += v.w; }
        }
        ((uint4*)sCnt[w])[lane] = tot;           // (the counters' storage, before it is zeroed for the ranking)
        ((uint4*)sPart[w])[lane] = bef;
    }
    __syncthreads();
    // exclusive scan of the column totals over the 256 digits (threads 0..255), tile bases
    {
        unsigned total = 0u, before = 0u;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < RS_WAVES; ++q) { total += sCnt[q][tid]; before += sPart[q][tid]; }
        }
        unsigned x = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = (unsigned)__shfl_up((int)x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) sScan[w] = x;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RS_WAVES; ++q)
            if (tid < 256) sCnt[q][tid] = 0u;
        unsigned wb = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) wb += (q < w) ? sScan[q] : 0u;
        if (tid < 256) sBase[tid] = wb + x - total + before;
        __syncthreads();
    }
    // wave-local stable ranks
    // (volatile: the counters are shared between the lanes of the wave, the compiler must not forward a lane's earlier load)
    unsigned rank[RS_ROUNDS];
    volatile unsigned* myCnt = sCnt[w];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool valid = (base + r * 64) < n;
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
            const unsigned prev = myCnt[dg];                                  // every peer reads the counter ...
            const unsigned below = (unsigned)__popcll(peers & d3f_lanemask_lt());
            rank[r] = prev + below;
            __builtin_amdgcn_wave_barrier();
            if (below == 0u) myCnt[dg] = prev + (unsigned)__popcll(peers);    // ... then the lowest peer advances it
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // Local sort through LDS, then coalesced stores.  Scattering straight from the registers made every store instruction hit
    // up to 64 different cache lines with 4 bytes each, and the eight XCD-private L2s each wrote their own bytes of every
    // line back: 84 us for one pass over 1.2 M items.  Ordered by digit inside the tile, the items of one digit are one
    // contiguous run in memory (32 items on average for an 8192-item tile = one 128-byte line).
    //   position inside the tile  p = locbase[digit] + (waves before)[digit] + rank;   memory position = p + delta[digit]
    {
        unsigned cnt = 0u;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < RS_WAVES; ++q) cnt += sCnt[q][tid];
        }
        unsigned x = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = (unsigned)__shfl_up((int)x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) sScan[w] = x;
        __syncthreads();
        if (tid < 256) {
            unsigned wb = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) wb += (q < w) ? sScan[q] : 0u;
            const unsigned loc = wb + x - cnt;            // first tile position of digit tid
            unsigned run = loc;
#pragma unroll
            for (int q = 0; q < RS_WAVES; ++q) {
                const unsigned c = sCnt[q][tid];
                sCnt[q][tid] = run;
                run += c;
            }
            sBase[tid] -= loc;                            // delta (mod 2^32)
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if ((base + r * 64) < n) {
            const unsigned dg = (key[r] >> shift) & 255u;
            const unsigned pl = myCnt[dg] + rank[r];
            sKey[pl] = key[r];
            sVal[pl] = val[r];
        }
    }
    __syncthreads();
    const int tile_n = min(RS_TILE, n - tile * RS_TILE);
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int pl = r * RS_THREADS + tid;
        if (pl < tile_n) {
            const unsigned k = sKey[pl];
            const unsigned dest = sBase[(k >> shift) & 255u] + (unsigned)pl;
            kdst[dest] = k;
            vdst[dest] = sVal[pl];
        }
    }
}
#endif

// Passes 1 .. RS_MAX_PASSES-1 histogram + every scatter; the caller has launched its key kernel (keys in key0, histogram of
// digit 0 in hist) before.  n_cap sizes the grids.
static inline int rs_sort_launch(const RsMeta* meta, int n_cap, unsigned* key0, unsigned* key1, unsigned* val0, unsigned* val1,
                                 unsigned* hist, hipStream_t stream) {
    const int tiles = rs_tiles(n_cap);
    static std::atomic<unsigned long long> lds_done{0ull};
    const void* const fns[] = {(const void*)rs_scatter_kernel};
    if (d3f_opt_in_lds(lds_done, fns, RS_SCATTER_LDS_BYTES) != D3F_OK) return D3F_ERR_HIP;
    for (int p = 0; p < RS_MAX_PASSES; ++p) {
        if (p > 0) {
            rs_hist_kernel<<<tiles, RS_THREADS, 0, stream>>>(meta, p, key0, key1, hist);
            D3F_LAUNCH_CHECK();
        }
        rs_scatter_kernel<<<tiles, RS_THREADS, RS_SCATTER_LDS_BYTES, stream>>>(meta, p, key0, key1, val0, val1, hist);
        D3F_LAUNCH_CHECK();
    }
    return D3F_OK;
}
