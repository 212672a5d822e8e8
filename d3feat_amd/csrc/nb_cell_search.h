// Full-list radius search, round 6: ONE WAVEFRONT PER QUERY for the distance tests, the stencil shared by the queries of a cell,
// then the rows of SIXTEEN queries ordered at once, four lanes per query, on 32-bit keys held in registers.
// (included by radius_neighbors.hip; semantics in its header: d2 = (dx*dx + dy*dy) + dz*dz in fp32 without FMA, strict d2 < r2,
// rows ascending by (d2, index) -- neighbors.cpp:125-208 / :211-332.)
//
// Why (profiles/r06_experiments.txt n1-n6): SQ counters of the 16-lanes-per-query kernel at level 0 (235 k queries) show 1105
// vector + 443 scalar instructions per wavefront of four queries with the vector pipe busy for the whole launch (a wave64 vector
// instruction occupies its SIMD for 4 cycles): the kernel is issue bound, not latency bound.  (a) every candidate slot of every query
// pays the 9-run lane -> run select chain (16 instructions + hazard nops), (b) the 64-key ordering network moves 64-bit
// (d2, index) keys across lanes: two cross-lane moves, a three-part compare and two selects per compare-exchange.  Here instead:
//   * a wavefront takes up to 16 consecutive queries; their cells, batch elements and coordinates are computed ONCE, one query per
//     lane, and then read lane by lane (v_readlane): everything about the current query is scalar;
//   * queries that ARE the supports come in cell order, so consecutive queries share their 27-cell stencil: its 9 runs are
//     resolved, mapped lane -> run and loaded into registers (6 candidates per lane: 384 per chunk) once per CELL; a query of the
//     same cell costs two or three 64-wide distance tests and nothing else.  (Queries in any other order reload per query.)
//   * hits are compacted by ballot + mbcnt into one {d2, index} array per wavefront in LDS and, when there are at most 64 (all but
//     ~0.1 % of the queries of a 3DMatch pyramid), copied to the query's own 64-entry list;
//   * ordering, after the last query: lane = (query, quarter); each lane loads 16 entries of its query's list and builds the keys
//     (d2 bits with the low 6 mantissa bits replaced by the entry's slot).  The 64 keys of a query are sorted by a bitonic network in
//     its mirrored form (every comparator puts the smaller key at the lower position: no direction flags): 18 of its 21 stages pair
//     registers of ONE lane -- v_min_u32 + v_max_u32, two instructions per comparator for sixteen queries at once -- and three
//     stages pair lanes of a quad through DPP.  27 vector instructions per query instead of ~150.
//     The truncation is checked, never trusted: if two neighbouring keys of a sorted list agree in the upper 26 bits of d2 (that
//     includes every exact tie) the query is ordered by exact rank counting over its list instead (~0.3 % of the queries) -- the
//     result is always the exact (d2, index) order.  Padding entries carry distinct keys above every real one.
//   * more than 64 hits (dense clouds): a 128-key network over the wavefront (two keys per lane), beyond that the same rank
//     counting, any count up to `cap`.
#pragma once

#define NBC_SLOTS 6                      // candidates per lane in registers
#define NBC_CHUNK (64 * NBC_SLOTS)       // candidates per chunk
#define NBC_QMAX 32                      // queries per wavefront (ordered in batches of eight, eight lanes each)

template <int J>
__device__ __forceinline__ unsigned nbc_xor_lane(unsigned v, int lane) {
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);  // row_ror:8 == lane ^ 8
    else if constexpr (J == 32) return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)v);
    else return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1F);                          // lane ^ J (bit mode)
}
// one compare-exchange stage of the network.  Element e = lane + 64 * slot; its partner is e ^ J, the merge it belongs to is
// ascending iff (e & K) == 0, and the lower element of a pair keeps the smaller key: for J < 64 the lanes that keep the smaller key
// are a compile-time pattern of the lane index -- a 64-bit constant that reaches v_cndmask as an SGPR pair (inverse ballot).
__host__ __device__ constexpr unsigned long long nbc_lane_bit(int bit) {      // lanes whose index has `bit` set
    unsigned long long m = 0ull;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)((l >> bit) & 1) << l;
    return m;
}
template <int K, int J, int SLOT>
__device__ __forceinline__ void nbc_stage(unsigned& k, int lane) {
    static_assert(J < 64, "distance 64 is the lane's own other key");
    constexpr int ij = J == 1 ? 0 : J == 2 ? 1 : J == 4 ? 2 : J == 8 ? 3 : J == 16 ? 4 : 5;
    constexpr int ik = K == 2 ? 1 : K == 4 ? 2 : K == 8 ? 3 : K == 16 ? 4 : 5;
    const unsigned p = nbc_xor_lane<J>(k, lane);
    const unsigned mn = min(k, p), mx = max(k, p);
    // lanes that keep the smaller key: (merge ascending) == (lower element of the pair).  The single-bit lane masks are constants
    // the compiler keeps or rematerialises as it likes; the stage's own pattern is ONE scalar instruction pinned here by `volatile`
    // (left to the compiler, the 21 + 54 patterns are hoisted out of the query loop: 100+ SGPRs, and the stencil's run offsets
    // went to scratch memory)
    unsigned long long keep_min;
    if constexpr (K >= 128 || (K == 64 && SLOT == 0))
        asm volatile("s_not_b64 %0, %1" : "=s"(keep_min) : "s"(nbc_lane_bit(ij)) : "scc");
    else if constexpr (K == 64)
        asm volatile("s_mov_b64 %0, %1" : "=s"(keep_min) : "s"(nbc_lane_bit(ij)));
    else
        asm volatile("s_xnor_b64 %0, %1, %2" : "=s"(keep_min) : "s"(nbc_lane_bit(ik)), "s"(nbc_lane_bit(ij)) : "scc");
    k = __builtin_amdgcn_inverse_ballot_w64(keep_min) ? mn : mx;
}
// 128 keys, two per lane (element lane + 64 * slot), ascending: k0 ends up holding elements 0..63
__device__ __forceinline__ void nbc_bitonic128(unsigned& k0, unsigned& k1, int lane) {
#define NBC_ST(K_, J_) do { nbc_stage<K_, J_, 0>(k0, lane); nbc_stage<K_, J_, 1>(k1, lane); } while (0)
    NBC_ST(2, 1);
    NBC_ST(4, 2); NBC_ST(4, 1);
    NBC_ST(8, 4); NBC_ST(8, 2); NBC_ST(8, 1);
    NBC_ST(16, 8); NBC_ST(16, 4); NBC_ST(16, 2); NBC_ST(16, 1);
    NBC_ST(32, 16); NBC_ST(32, 8); NBC_ST(32, 4); NBC_ST(32, 2); NBC_ST(32, 1);
    NBC_ST(64, 32); NBC_ST(64, 16); NBC_ST(64, 8); NBC_ST(64, 4); NBC_ST(64, 2); NBC_ST(64, 1);   // slot 1 descending
    { const unsigned a = min(k0, k1), b = max(k0, k1); k0 = a; k1 = b; }                             // K = 128, J = 64
    NBC_ST(128, 32); NBC_ST(128, 16); NBC_ST(128, 8); NBC_ST(128, 4); NBC_ST(128, 2); NBC_ST(128, 1);
#undef NBC_ST
}

__device__ __forceinline__ int nbc_rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float nbc_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// ---- the 64 keys of a query in the registers of eight lanes: element e = 8 * part + r (part = lane & 7, r = register) --------
// Mirrored bitonic network: the first step of the merge of block size K pairs e with e ^ (K - 1), the following steps e with e ^ J
// (J = K / 4 ... 1); every comparator leaves the smaller key at the lower position.  15 of the 21 steps pair registers of one lane
// (v_min_u32 + v_max_u32 per comparator, eight queries at once), six pair lanes of the group through DPP.
#define NBC_GROUP 8                      // lanes per query in the ordering phase
#define NBC_BATCH 8                      // queries ordered at once (64 / NBC_GROUP)
#define NBC_CE(A_, B_) do { const unsigned lo_ = min(k[A_], k[B_]), hi_ = max(k[A_], k[B_]); k[A_] = lo_; k[B_] = hi_; } while (0)
template <int K>
__device__ __forceinline__ void nbc_g_mirror(unsigned (&k)[8]) {            // K <= 8: both elements in this lane
#pragma unroll
    for (int base = 0; base < 8; base += K)
#pragma unroll
        for (int t = 0; t < K / 2; ++t) NBC_CE(base + t, base + K - 1 - t);
}
template <int J>
__device__ __forceinline__ void nbc_g_clean(unsigned (&k)[8]) {             // J <= 4
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r & J) == 0) NBC_CE(r, r | J);
}
// partner in another lane of the group: lane ^ PX, register MIRROR ? 7 - r : r; the lane with the lower part keeps the smaller key
template <int PX, bool MIRROR>
__device__ __forceinline__ void nbc_g_cross(unsigned (&k)[8]) {
    // lane ^ 1: quad_perm [1,0,3,2]; ^ 2: quad_perm [2,3,0,1]; ^ 3: quad_perm [3,2,1,0]; ^ 7: row_half_mirror
    constexpr int ctrl = PX == 1 ? 0xB1 : PX == 2 ? 0x4E : PX == 3 ? 0x1B : 0x141;
    // lanes whose part has the highest bit of PX clear
    constexpr unsigned long long lower = PX == 1 ? 0x5555555555555555ull : (PX == 2 || PX == 3) ? 0x3333333333333333ull : 0x0F0F0F0F0F0F0F0Full;
    unsigned nk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned p = (unsigned)__builtin_amdgcn_mov_dpp((int)k[MIRROR ? 7 - r : r], ctrl, 0xF, 0xF, true);
        const unsigned mn = min(k[r], p), mx = max(k[r], p);
        nk[r] = __builtin_amdgcn_inverse_ballot_w64(lower) ? mn : mx;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) k[r] = nk[r];
}
__device__ __forceinline__ void nbc_group_sort64(unsigned (&k)[8]) {
    nbc_g_mirror<2>(k);
    nbc_g_mirror<4>(k); nbc_g_clean<1>(k);
    nbc_g_mirror<8>(k); nbc_g_clean<2>(k); nbc_g_clean<1>(k);
    nbc_g_cross<1, true>(k);                                                 // K = 16: e ^ 15
    nbc_g_clean<4>(k); nbc_g_clean<2>(k); nbc_g_clean<1>(k);
    nbc_g_cross<3, true>(k);                                                 // K = 32: e ^ 31
    nbc_g_cross<1, false>(k);                                                // J = 8
    nbc_g_clean<4>(k); nbc_g_clean<2>(k); nbc_g_clean<1>(k);
    nbc_g_cross<7, true>(k);                                                 // K = 64: e ^ 63
    nbc_g_cross<2, false>(k);                                                // J = 16
    nbc_g_cross<1, false>(k);                                                // J = 8
    nbc_g_clean<4>(k); nbc_g_clean<2>(k); nbc_g_clean<1>(k);
}
#undef NBC_CE

template <bool SORTED_Q, bool INTERNAL = false>   // INTERNAL: `inv` given (compile-time: the two numberings do not share registers)
__global__ void __launch_bounds__(256, 5)   // five workgroups per CU (96 registers; the INTERNAL instance would take 99 = four)
nb_cell_search_kernel(const float* __restrict__ q, int Nq, const int* __restrict__ qlens, int B,
                      const NbElem* __restrict__ el, const int* __restrict__ cell_start, const int* __restrict__ cell_base,
                      const float4* __restrict__ sorted, float r2, int pad, const int* __restrict__ ns_dev,
                      int* __restrict__ out, int ld, int width, int cap, int* __restrict__ status, int want_kmax, int Q, int dbg_arg,
                      unsigned long long* __restrict__ prof_arg, const int* __restrict__ inv) {
    // measurement aids (phase skipping, per-wavefront phase clocks: tools/ubench/nbc_phase_profile.py) exist in -DD3F_NBC_MEASURE builds
    // only: as run-time arguments they cost the production kernel scalar registers and a handful of scalar tests per query
#ifdef D3F_NBC_MEASURE
    const int dbg = dbg_arg;
    unsigned long long* const prof = prof_arg;
#else
    constexpr int dbg = 0;
    constexpr unsigned long long* prof = nullptr;
    (void)dbg_arg; (void)prof_arg;
#endif
    // inv != NULL: the INTERNAL numbering -- row j of `out` is the j-th query in cell order and the entries are positions in the
    // cell-sorted support arrays (inv[index]); the order of a row is the reference's all the same (ties by the ORIGINAL index)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ NbElem sel[NB_EL_LDS];
    __shared__ int sOff[NB_EL_LDS + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t_start = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // per wavefront: the hits of the current query {d2 bits, support index} [cap], then one 64-entry list of d2 bits and one of
    // indices for each query of the batch being collected [8][64]
    char* wbase = smem + (size_t)wave * ((size_t)cap * 8 + NBC_BATCH * 64 * 8);
    uint2* hk = (uint2*)wbase;
    unsigned* ld2 = (unsigned*)(wbase + (size_t)cap * 8);
    int* lidx = (int*)(ld2 + NBC_BATCH * 64);
    const bool staged = B <= NB_EL_LDS;
    if (staged) {
        constexpr int W = (int)(sizeof(NbElem) / sizeof(int));
        for (int i = threadIdx.x; i < B * W; i += blockDim.x) ((int*)sel)[i] = ((const int*)el)[i];
        if (threadIdx.x <= B) {
            int s = 0;
            for (int j = 0; j < (int)threadIdx.x; ++j) s += qlens[j];
            sOff[threadIdx.x] = s;
        }
        __syncthreads();
    }
    int nq_real = 0;
    for (int j = 0; j < B; ++j) nq_real += qlens[j];
    const int nq = min(Nq, nq_real);                      // Nq is the capacity, sum(qlens) the real number of queries
    const int nw = (nq + Q - 1) / Q;                      // wavefronts that have queries
    const int nblk = (nw + 3) >> 2;
    if ((int)blockIdx.x >= nblk) return;
    // one contiguous run of query blocks per XCD (common.h): neighbouring blocks read the same candidate runs
    const int wg = (int)d3f_xcd_tile(blockIdx.x, (unsigned)nblk) * 4 + wave;
    if (wg >= nw) return;
    const int p0 = wg * Q, cnt = min(Q, nq - p0);
    if (pad == D3F_PAD_NUM_SUPPORTS) pad = *ns_dev;

    // ---- prologue, one query per lane: coordinates, index, batch element, cell ------------------------------------------
    float vqx = 0.f, vqy = 0.f, vqz = 0.f;
    int vqi = 0, vb = 0, vcx = 0, vcy = 0, vcz = 0;
    if (lane < cnt) {
        const int p = p0 + lane;
        if (SORTED_Q) {
            const float4 me = sorted[p];          // position AND index of the p-th support in cell order: one 16-byte record
            vqx = me.x; vqy = me.y; vqz = me.z; vqi = __float_as_int(me.w);
        } else {
            vqx = q[3 * (size_t)p]; vqy = q[3 * (size_t)p + 1]; vqz = q[3 * (size_t)p + 2]; vqi = p;
        }
        // batch element: the last one starting at or before the query (cell-sorted supports are grouped by element as well:
        // an element's cells are one contiguous range, so position and index give the same element)
        if (staged) {
            for (int j = 1; j < B; ++j) vb = (p >= sOff[j]) ? j : vb;
        } else {
            for (int j = 1, start = qlens[0]; j < B; ++j) { if (p >= start) vb = j; start += qlens[j]; }
        }
        const NbElem e = staged ? sel[vb] : el[vb];
        nb_cell_of(e, vqx, vqy, vqz, vcx, vcy, vcz);
        vcx = min(max(vcx, -2), e.dims[0] + 1);
        vcy = min(max(vcy, -2), e.dims[1] + 1);
        vcz = min(max(vcz, -2), e.dims[2] + 1);
    }

    // internal numbering: the hit records hold POSITIONS in the cell-sorted array; the original index (the reference's tie-break,
    // read in the rare exact paths only when two d2 are bit-equal) is one gather away
#define NBC_ORIG(Y_) (INTERNAL ? __float_as_int(sorted[Y_].w) : (Y_))
    float4 cand[NBC_SLOTS];
    int cb = -1, ccx = 0, ccy = 0, ccz = 0;     // the stencil in registers: batch element and cell (wave-uniform)
    int T = 0;                                   // candidates of the stencil
    // run prefix / run offset, wave-uniform.  Named scalars, not arrays: an array indexed in the select chain below is kept in
    // scratch memory by the compiler (it rewrites the chain as "select the index, then load")
    int pre1 = 0, pre2 = 0, pre3 = 0, pre4 = 0, pre5 = 0, pre6 = 0, pre7 = 0, pre8 = 0;
    int off0 = 0, off1 = 0, off2 = 0, off3 = 0, off4 = 0, off5 = 0, off6 = 0, off7 = 0, off8 = 0;
    int nmax = 0;
    bool over = false;
    int vn = -1;                                 // lane i: hits of query i when its row is still to be ordered from its list (<= 64)

    // candidates [c0, c0 + NBC_CHUNK) of the current stencil -> cand[]; a lane beyond the list holds a point at 3e38 (d2 = +inf)
#define NBC_LOAD_CHUNK(C0_)                                                                                    \
    do {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < NBC_SLOTS; ++u) {                                                \
            cand[u] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);                                             \
            if ((C0_) + u * 64 < T) { /* wave-uniform */                                                       \
                const int v = (C0_) + u * 64 + lane;                                                           \
                int o = off0;                                                                                  \
                o = (v >= pre1) ? off1 : o; o = (v >= pre2) ? off2 : o; o = (v >= pre3) ? off3 : o;            \
                o = (v >= pre4) ? off4 : o; o = (v >= pre5) ? off5 : o; o = (v >= pre6) ? off6 : o;            \
                o = (v >= pre7) ? off7 : o; o = (v >= pre8) ? off8 : o;                                        \
                if (v < T) {                                                                                   \
                    /* internal numbering: the record's POSITION is what the rows hold (its index only breaks ties): 12 bytes */ \
                    /* are loaded and the fourth register is written at once -- overwriting .w of a 16-byte load would keep  */ \
                    /* the position in one more register per slot until the load returns (99 registers = four waves)        */ \
                    if (INTERNAL) {                                                                            \
                        const float* sp_ = (const float*)&sorted[v + o];                                       \
                        cand[u].x = sp_[0]; cand[u].y = sp_[1]; cand[u].z = sp_[2];                            \
                        cand[u].w = __int_as_float(v + o);                                                     \
                    } else cand[u] = sorted[v + o];                                                            \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

    // measurement aid (prof != NULL): shader-clock cycles per phase, one record per wavefront
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = 0;
#define NBC_T0() do { if (prof) t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define NBC_T1(P_) do { if (prof) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tp[P_] += t1_ - t0; t0 = t1_; } } while (0)
    if (prof) { tp[0] = __builtin_amdgcn_s_memtime() - t_start; }
    NBC_T0();
    for (int i = 0; i < cnt; ++i) {
        const int b = nbc_rl(vb, i), cx = nbc_rl(vcx, i), cy = nbc_rl(vcy, i), cz = nbc_rl(vcz, i);
        const float qx = nbc_rlf(vqx, i), qy = nbc_rlf(vqy, i), qz = nbc_rlf(vqz, i);
        if (b != cb || cx != ccx || cy != ccy || cz != ccz || (dbg & 8)) {
            cb = b; ccx = cx; ccy = cy; ccz = cz;
            // lanes 0..8: start and length of the 9 (y, z) rows of the stencil (x-adjacent cells are contiguous)
            int dimx, dimy, dimz, cbase;
            if (staged) { dimx = sel[b].dims[0]; dimy = sel[b].dims[1]; dimz = sel[b].dims[2]; cbase = sel[b].cbase; }
            else { dimx = el[b].dims[0]; dimy = el[b].dims[1]; dimz = el[b].dims[2]; cbase = el[b].cbase; }
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, dimx - 1);
            int bound = 0, len_l = 0;
            if (lane < 9) {
                const int y = cy + (lane % 3) - 1, z = cz + (lane / 3) - 1;
                if (x0 <= x1 && y >= 0 && y < dimy && z >= 0 && z < dimz) {
                    const int rowbase = cbase + dimx * (y + dimy * z);
                    bound = d3f_scan_at(cell_start, cell_base, rowbase + x0);
                    len_l = d3f_scan_at(cell_start, cell_base, rowbase + x1 + 1) - bound;
                }
            }
            int acc = 0;
#define NBC_RUN(J_, PRE_, OFF_) { PRE_ = acc; OFF_ = nbc_rl(bound, J_) - acc; acc += nbc_rl(len_l, J_); }
            int pre0;
            NBC_RUN(0, pre0, off0) NBC_RUN(1, pre1, off1) NBC_RUN(2, pre2, off2) NBC_RUN(3, pre3, off3) NBC_RUN(4, pre4, off4)
            NBC_RUN(5, pre5, off5) NBC_RUN(6, pre6, off6) NBC_RUN(7, pre7, off7) NBC_RUN(8, pre8, off8)
#undef NBC_RUN
            (void)pre0;
            T = acc;
            if (T <= NBC_CHUNK) NBC_LOAD_CHUNK(0);
        }
        NBC_T1(1);
        // ---- distance tests: hits -> LDS ---------------------------------------------------------------------------------
        int n = 0;
        for (int c0 = 0; c0 < ((dbg & 16) ? 0 : T); c0 += NBC_CHUNK) {
            if (T > NBC_CHUNK) NBC_LOAD_CHUNK(c0);  // a stencil beyond the registers is streamed per query (dense clouds only)
#pragma unroll
            for (int u = 0; u < NBC_SLOTS; ++u) {
                if (c0 + u * 64 >= T) break;        // wave-uniform
                const float dx = __fsub_rn(qx, cand[u].x), dy = __fsub_rn(qy, cand[u].y), dz = __fsub_rn(qz, cand[u].z);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const bool hit = d2 < r2;
                const unsigned long long m = __ballot(hit);
                if (hit) {
                    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (pos < cap) hk[pos] = make_uint2(__float_as_uint(d2), (unsigned)__float_as_int(cand[u].w));
                }
                n += __popcll(m);
            }
        }
        NBC_T1(2);
        nmax = max(nmax, n);
        over = over || (n > cap);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n <= 64) {
            // the query's own list: its hits, then padding entries whose keys are distinct and above every real key (finite
            // d2 bits are below 0x7F800000); ordered after the last query, sixteen queries at a time
            uint2 h = make_uint2(0x80000000u | ((unsigned)lane << 6), 0u);
            if (lane < n) h = hk[lane];
            ld2[(i & (NBC_BATCH - 1)) * 64 + lane] = h.x;
            lidx[(i & (NBC_BATCH - 1)) * 64 + lane] = (int)h.y;
            vn = (lane == i) ? n : vn;
        } else {
            // ---- dense neighbourhood (rare): ordered now, by the whole wavefront, from the hit array ----------------------
            const int m = min(n, cap);
            int* row = out + (size_t)(INTERNAL ? p0 + i : nbc_rl(vqi, i)) * ld;
            bool need_exact = true;
            if (n <= 128 && n <= cap && width < 64) {
                // 128 keys, two per lane (upper 25 bits of d2 | slot); the row is the head of elements 0..63
                unsigned k0 = (hk[lane].x & 0xFFFFFF80u) | (unsigned)lane, k1 = 0xFFFFFFFFu;
                if (lane + 64 < m) k1 = (hk[lane + 64].x & 0xFFFFFF80u) | (unsigned)(lane + 64);
                nbc_bitonic128(k0, k1, lane);
                const unsigned kn = (unsigned)__shfl_down((int)k0, 1);
                const bool clash = (lane < width) && ((k0 >> 7) == (kn >> 7));   // (width < 64 <= m: lane + 1 is a real element)
                need_exact = __ballot(clash) != 0ull;
                if (!need_exact && lane < width) row[lane] = (int)hk[k0 & 127u].y;
            }
            if (need_exact) {
                // exact rank counting over the hit array: rank = number of hits with a smaller (d2, index) key (keys are unique)
                for (int e0 = 0; e0 < m; e0 += 64) {
                    const int ei = e0 + lane;
                    const uint2 mine = hk[ei < m ? ei : 0];
                    int rank = 0;
                    if (INTERNAL) {
                        // (bit-equal d2 only: the gather of the original indices stays out of the common iteration and out of the
                        // register budget of the kernel)
#pragma unroll 1
                        for (int j = 0; j < m; ++j) {
                            const uint2 o = hk[j];
                            bool before = o.x < mine.x;
                            if (o.x == mine.x && o.y != mine.y) before = NBC_ORIG((int)o.y) < NBC_ORIG((int)mine.y);
                            rank += before ? 1 : 0;
                        }
                    } else {
                        for (int j = 0; j < m; ++j) {
                            const uint2 o = hk[j];
                            rank += (o.x < mine.x || (o.x == mine.x && (int)o.y < (int)mine.y)) ? 1 : 0;
                        }
                    }
                    if (ei < m && rank < width) row[rank] = (int)mine.y;
                }
            }
            for (int j = m + lane; j < width; j += 64) row[j] = pad;
        }
        __builtin_amdgcn_wave_barrier();            // the next query's hits overwrite the array
        NBC_T1(3);
        if ((i & (NBC_BATCH - 1)) != NBC_BATCH - 1 && i != cnt - 1) continue;
        // ---- ordering of the batch's rows: lane = (query of the batch, eighth of its list) --------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int i0 = i & ~(NBC_BATCH - 1);         // first query of the batch
        const int qd = lane >> 3, part = lane & 7;
        const int nqd = __shfl(vn, i0 + qd);                                // hits of this lane's query (-1: nothing to do)
        const int qiq = INTERNAL ? p0 + i0 + qd : __shfl(vqi, i0 + qd);          // its row
        unsigned k[8];
        {
            const uint4* src = (const uint4*)(ld2 + qd * 64 + part * 8);
            const uint4 v0 = src[0], v1 = src[1];
            k[0] = (v0.x & 0xFFFFFFC0u) | (unsigned)(part * 8 + 0); k[1] = (v0.y & 0xFFFFFFC0u) | (unsigned)(part * 8 + 1);
            k[2] = (v0.z & 0xFFFFFFC0u) | (unsigned)(part * 8 + 2); k[3] = (v0.w & 0xFFFFFFC0u) | (unsigned)(part * 8 + 3);
            k[4] = (v1.x & 0xFFFFFFC0u) | (unsigned)(part * 8 + 4); k[5] = (v1.y & 0xFFFFFFC0u) | (unsigned)(part * 8 + 5);
            k[6] = (v1.z & 0xFFFFFFC0u) | (unsigned)(part * 8 + 6); k[7] = (v1.w & 0xFFFFFFC0u) | (unsigned)(part * 8 + 7);
        }
        if (!(dbg & 1)) nbc_group_sort64(k);
        // neighbouring keys that agree in the upper 26 bits of d2 (two hits within 2^-17 of each other, or tied): the exact path
        // decides that query.  (k[7] against the next lane's k[0]; the last lane of a group has no successor)
        bool clash = false;
#pragma unroll
        for (int r = 0; r < 7; ++r) clash = clash || ((k[r] ^ k[r + 1]) < 64u);
        const unsigned nx = (unsigned)__builtin_amdgcn_mov_dpp((int)k[0], 0x101, 0xF, 0xF, true);    // row_shl:1: lane + 1's k[0]
        clash = clash || (part != 7 && (k[7] ^ nx) < 64u);
        if (dbg & 2) clash = false;
        const bool live = nqd >= 0 && i0 + qd <= i;
        const unsigned long long cm = __ballot(clash && live);
        const bool redo = ((cm >> (lane & 56)) & 0xFFull) != 0ull;       // some lane of this group saw a clash
        if (live && !redo) {
            int* row = out + (size_t)qiq * ld;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int e = part * 8 + r;
                if (e < width) {
                    int v = pad;
                    if (e < nqd) v = lidx[qd * 64 + (int)(k[r] & 63u)];
                    row[e] = v;
                }
            }
            for (int e = 64 + part; e < width; e += 8) row[e] = pad;         // (rows wider than the list: padding only)
        }
        NBC_T1(4);
        // ---- exact rank counting over the list of every query that saw a clash (whole wavefront per query) ------------------
        unsigned long long todo = cm;
        while (todo && !(dbg & 4)) {
            const int g = (int)(__builtin_ctzll(todo) >> 3);
            todo &= ~(0xFFull << (8 * g));
            const int m = nbc_rl(vn, i0 + g);
            int* row = out + (size_t)(INTERNAL ? p0 + i0 + g : nbc_rl(vqi, i0 + g)) * ld;
            const unsigned md = ld2[g * 64 + (lane < m ? lane : 0)];
            const int mi = lidx[g * 64 + (lane < m ? lane : 0)];
            int rank = 0;
            // ties by the ORIGINAL index: lane j holds element j's (one gather per lane in the internal numbering, then lane reads)
            const int mo = NBC_ORIG(mi);
            for (int j = 0; j < m; ++j) {
                const unsigned od = ld2[g * 64 + j];
                const int oo = INTERNAL ? __builtin_amdgcn_readlane(mo, j) : lidx[g * 64 + j];
                rank += (od < md || (od == md && oo < mo)) ? 1 : 0;
            }
            if (lane < m && rank < width) row[rank] = mi;
            for (int j = m + lane; j < width; j += 64) row[j] = pad;
        }
        // the batch is done: its queries must not be ordered again with the next batch's lists
        vn = (lane >= i0 && lane <= i) ? -1 : vn;
        __builtin_amdgcn_wave_barrier();
    }
#undef NBC_LOAD_CHUNK
#undef NBC_ORIG
    NBC_T1(5);
    if (prof && lane == 0) {        // one record per wavefront (no atomics: they would serialise the launch being measured)
        for (int k = 0; k < 6; ++k) prof[(size_t)wg * 8 + k] = tp[k];
        prof[(size_t)wg * 8 + 6] = t_start;
        prof[(size_t)wg * 8 + 7] = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0) {
        // one shared word: read first, update only when this wavefront raises the maximum (a handful of times per launch)
        if (want_kmax && nmax > __hip_atomic_load(&status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&status[0], nmax);
        if (over) atomicOr(&status[1], D3F_ST_HIT_OVERFLOW);
    }
}
