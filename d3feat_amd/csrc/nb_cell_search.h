// Full-list radius search, round 6: ONE WAVEFRONT PER QUERY, the stencil shared by the queries of a cell, ordering on 32-bit keys.
// (included by radius_neighbors.hip; semantics in its header: d2 = (dx*dx + dy*dy) + dz*dz in fp32 without FMA, strict d2 < r2,
// rows ascending by (d2, index) -- neighbors.cpp:125-208 / :211-332.)
//
// Why (profiles/r06_experiments.txt n1): SQ counters of the 16-lanes-per-query kernel at level 0 (235 k queries) show 921 vector
// + 378 scalar instructions per wavefront of four queries and the vector pipe busy for the whole launch -- the kernel is issue
// bound, not latency bound: (a) every candidate slot of every query pays the 9-run lane -> run select chain (16 instructions +
// hazard nops), (b) the 64-key ordering network moves 64-bit (d2, index) keys: two cross-lane moves, a three-part compare and two
// selects per compare-exchange, four keys per lane.  Here instead:
//   * a wavefront takes Q consecutive queries; the queries' cells, batch elements and coordinates are computed ONCE for the
//     wavefront, one query per lane, and then read lane by lane (v_readlane): everything about the current query is scalar;
//   * queries that ARE the supports come in cell order, so consecutive queries share their 27-cell stencil: its 9 runs are
//     resolved, mapped lane -> run and loaded into registers (4 candidates per lane: 256 per chunk) once per CELL; a query of the
//     same cell costs two 64-wide distance tests and nothing else.  (Queries in any other order reload per query.)
//   * hits are compacted by ballot + mbcnt into one {d2, index} array per wavefront in LDS;
//   * ordering: key = (d2 bits with the low 6 mantissa bits replaced by the hit's slot): ONE 32-bit key per lane, a 64-lane
//     bitonic network of v_min_u32 / v_max_u32 (partners through DPP for distances 1, 2, 8, ds_swizzle for 4, 16, ds_bpermute for
//     32): 3 vector instructions per stage instead of ~36.  The truncation is checked, never trusted: if two hits of a query agree
//     in the upper 26 bits of d2 (that includes every exact tie) the query is ordered by exact rank counting over the LDS array
//     instead (about 0.2 % of the queries of a 3DMatch level) -- the result is always the exact (d2, index) order.
//   * more than 64 hits (dense clouds): the same rank counting, any count up to `cap`.
#pragma once

#define NBC_SLOTS 4                      // candidates per lane in registers
#define NBC_CHUNK (64 * NBC_SLOTS)       // candidates per chunk
#define NBC_QMAX 32                      // queries per wavefront (one per lane in the prologue)

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
// 64 keys, one per lane, ascending over the lanes
__device__ __forceinline__ void nbc_bitonic64(unsigned& k, int lane) {
#define NBC_ST(K_, J_) nbc_stage<K_, J_, 0>(k, lane)
    NBC_ST(2, 1);
    NBC_ST(4, 2); NBC_ST(4, 1);
    NBC_ST(8, 4); NBC_ST(8, 2); NBC_ST(8, 1);
    NBC_ST(16, 8); NBC_ST(16, 4); NBC_ST(16, 2); NBC_ST(16, 1);
    NBC_ST(32, 16); NBC_ST(32, 8); NBC_ST(32, 4); NBC_ST(32, 2); NBC_ST(32, 1);
    NBC_ST(64, 32); NBC_ST(64, 16); NBC_ST(64, 8); NBC_ST(64, 4); NBC_ST(64, 2); NBC_ST(64, 1);
#undef NBC_ST
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

template <bool SORTED_Q>
__global__ void __launch_bounds__(256)
nb_cell_search_kernel(const float* __restrict__ q, int Nq, const int* __restrict__ qlens, int B,
                      const NbElem* __restrict__ el, const int* __restrict__ cell_start, const int* __restrict__ cell_base,
                      const float4* __restrict__ sorted, float r2, int pad, const int* __restrict__ ns_dev,
                      int* __restrict__ out, int ld, int width, int cap, int* __restrict__ status, int want_kmax, int Q, int dbg,
                      unsigned long long* __restrict__ prof) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ NbElem sel[NB_EL_LDS];
    __shared__ int sOff[NB_EL_LDS + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t_start = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // per wavefront: the hits of the current query {d2 bits, support index} [cap], then the finished rows of its Q queries
    // [Q][width]: rows are written to memory after the last query -- a store inside the query loop would be waited for by the
    // next query's `s_waitcnt vmcnt(0)` (loads and stores retire through one counter): a write round trip per query
    const int wave_words = (2 * cap + Q * width + 1) & ~1;      // (8-byte aligned hit records)
    uint2* hk = (uint2*)((int*)smem + (size_t)wave * wave_words);
    int* orow = (int*)(hk + cap);
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

    float4 cand[NBC_SLOTS];
    int cb = -1, ccx = 0, ccy = 0, ccz = 0;     // the stencil in registers: batch element and cell (wave-uniform)
    int T = 0;                                   // candidates of the stencil
    // run prefix / run offset, wave-uniform.  Named scalars, not arrays: an array indexed in the select chain below is kept in
    // scratch memory by the compiler (it rewrites the chain as "select the index, then load")
    int pre1 = 0, pre2 = 0, pre3 = 0, pre4 = 0, pre5 = 0, pre6 = 0, pre7 = 0, pre8 = 0;
    int off0 = 0, off1 = 0, off2 = 0, off3 = 0, off4 = 0, off5 = 0, off6 = 0, off7 = 0, off8 = 0;
    int nmax = 0;
    bool over = false;

    // candidates [c0, c0 + 256) of the current stencil -> cand[]; a lane beyond the list holds a point at 3e38 (its d2 is +inf)
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
                if (v < T) cand[u] = sorted[v + o];                                                            \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

    // measurement aid (prof != NULL): shader-clock cycles per phase, summed and maximised over the wavefronts
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
        const int m = min(n, cap);
        int* row = orow + i * width;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool need_exact = true;
        if (n <= 64) {
            // one 32-bit key per lane: upper 26 bits of d2 | slot; padding keys sort last
            unsigned k = 0xFFFFFFFFu;
            if (lane < m) k = (hk[lane].x & 0xFFFFFFC0u) | (unsigned)lane;
            if (!(dbg & 1)) nbc_bitonic64(k, lane);
            // two hits of the row within 2^-17 of each other (or tied): the exact path decides
            const unsigned kn = (unsigned)__shfl_down((int)k, 1);
            const bool clash = (lane + 1 < m) && (lane < width) && ((k >> 6) == (kn >> 6));
            need_exact = !(dbg & 2) && __ballot(clash) != 0ull;
            if (!need_exact) {
                const int mw = min(m, width);
                if (lane < mw) row[lane] = (int)hk[k & 63u].y;
            }
        } else if (n <= 128 && n <= cap && width < 64) {
            // dense neighbourhoods: 128 keys, two per lane (upper 25 bits of d2 | slot); the row is the head of elements 0..63
            unsigned k0 = (hk[lane].x & 0xFFFFFF80u) | (unsigned)lane, k1 = 0xFFFFFFFFu;
            if (lane + 64 < m) k1 = (hk[lane + 64].x & 0xFFFFFF80u) | (unsigned)(lane + 64);
            if (!(dbg & 1)) nbc_bitonic128(k0, k1, lane);
            const unsigned kn = (unsigned)__shfl_down((int)k0, 1);
            const bool clash = (lane < width) && ((k0 >> 7) == (kn >> 7));       // (width < 64 <= m: lane + 1 is a real element)
            need_exact = !(dbg & 2) && __ballot(clash) != 0ull;
            if (!need_exact && lane < width) row[lane] = (int)hk[k0 & 127u].y;
        }
        NBC_T1(3);
        if (need_exact && !(dbg & 4)) {
            if ((dbg & 32) && lane == 0) { atomicAdd(&status[0], 1); atomicMax(&status[1], m); }
            // exact rank counting over the LDS array: rank = number of hits with a smaller (d2, index) key (keys are unique)
            for (int e0 = 0; e0 < m; e0 += 64) {
                const int ei = e0 + lane;
                const uint2 mine = hk[ei < m ? ei : 0];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const uint2 o = hk[j];
                    rank += (o.x < mine.x || (o.x == mine.x && (int)o.y < (int)mine.y)) ? 1 : 0;
                }
                if (ei < m && rank < width) row[rank] = (int)mine.y;
            }
        }
        for (int j = m + lane; j < width; j += 64) row[j] = pad;
        __builtin_amdgcn_wave_barrier();            // the next query's hits overwrite the array
        NBC_T1(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < cnt; ++i) {
        int* grow = out + (size_t)nbc_rl(vqi, i) * ld;
        for (int j = lane; j < width; j += 64) grow[j] = orow[i * width + j];
    }
#undef NBC_LOAD_CHUNK
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
