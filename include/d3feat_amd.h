/*
 * d3feat_amd -- C ABI of the MI355X-native D3Feat inference hot path (libd3feat_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / HIP types in the signatures.
 * Every pointer named *_dev (and every tensor argument) is a DEVICE pointer in HBM unless stated
 * otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream).  All entry points
 * are asynchronous on `stream`, allocate nothing (the caller owns outputs and the workspace) and
 * return D3F_OK or a negative D3F_ERR_* code.  Data-dependent failures that can only be detected
 * on the device are reported through a caller-provided int status[] in HBM (see each function).
 *
 * Each function cites the reference interface (paths relative to the XuyangBai/D3Feat checkout)
 * it replaces.  INTEGRATION.md shows the binding a maintainer of the reference would add.
 */
#ifndef D3FEAT_AMD_H
#define D3FEAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3F_OK 0
#define D3F_ERR_HIP (-1)        /* a HIP runtime call / kernel launch failed */
#define D3F_ERR_WORKSPACE (-2)  /* workspace too small (see *_workspace_bytes) */
#define D3F_ERR_ARG (-3)        /* invalid argument (negative size, B out of range, bad leading dimension ...) */

/* bits of the device-side status word */
#define D3F_ST_EMPTY_ELEMENT 1   /* a batch element has zero points (UB in the reference, cloud.cpp:30,51)  */
#define D3F_ST_NEG_CELL 2        /* floor((p-origin)/dl) < 0 (the reference's (size_t) cast would be UB, :52-54) */
#define D3F_ST_KEY_RANGE 4       /* voxel key >= 2^56 */
#define D3F_ST_HIT_OVERFLOW 8    /* a query has more in-radius supports than D3F_NEIGHBOR_CAP */
#define D3F_ST_OUT_OVERFLOW 16   /* capacity mode: more output rows than the caller's buffer holds (nothing is written out of bounds) */
#define D3F_ST_KEY_WIDTH 32      /* capacity mode of the grid subsampling: (element, voxel key) needs more than the 32 bits of the
                                    stage-0 sort key (a grid of > 2^32 / B cells): empty result, the synchronous call handles it */

/* pad_value of the neighbour searches meaning "the number of supports as known on the DEVICE" (sum of s_lens_dev) --
 * what BatchOrderedNeighbors pads with (neighbors.cpp:324) when the host only knows an upper bound of Ns */
#define D3F_NB_NO_KMAX 2
#define D3F_PAD_NUM_SUPPORTS (-2147483647 - 1)

#define D3F_MAX_BATCH 255        /* batch elements per stacked call */
#define D3F_NEIGHBOR_CAP 1024    /* max in-radius supports per query that can be ordered */
#define D3F_NUM_KP_MAX 16        /* kernel points per KPConv (reference uses 15) */

int d3f_version(void);

/* Measurement aid (no reference counterpart): launches the one-thread kernel `d3f_trace_marker_kernel` on `stream`, so that
 * a kernel trace of a run can be cut at the ends of a timed region (bench.py; tools/rocpd_summary.py --timed-region). */
int d3f_trace_marker(int id, void* stream);
/* Capacity mode (no reference counterpart: the reference's ops return their sizes to the host one by one): packs up to four device
 * int blocks (sizes, status words, lens) into dst[0 .. na+nb+nc+nd) in ONE launch, then zeroes the nclear words
 * clear[clear_first + i * clear_step] -- the sticky flag words of the searches ([kmax | flags] pairs: first 1, step 2), ready for the
 * next replay (`clear` may alias a source).  Any pointer whose count is 0 may be NULL. */
int d3f_pack_status(int* dst, const int* a, int na, const int* b, int nb, const int* c, int nc, const int* d, int nd,
                    int* clear, int nclear, int clear_first, int clear_step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Device-resident sizes ("capacity mode").  The reference's ops have data-dependent output sizes, which costs a host
 * round trip per op (5 per fragment on this path).  Every entry point below therefore also works with sizes that live
 * in HBM: the int size arguments are then UPPER BOUNDS (they size launch grids and buffers) and the real sizes are read on
 * the device -- from the lens arrays for the preprocessing ops, from the optional `*_dev` pointers (NULL = use the host
 * value) for the network ops.  A whole fragment is then a fixed launch sequence: captured once as a HIP graph, replayed
 * for any cloud up to the capacity, with one status read-back at the end.
 * ------------------------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------------
 * Grid subsampling.
 * Replaces tf_custom_ops/tf_subsampling: BatchGridSubsampling (tf_batch_subsampling.cpp:8-20,
 * kernel :26-123 -> grid_subsampling/grid_subsampling.cpp:101-149) and GridSubsampling
 * (tf_subsampling.cpp:8-11 -> grid_subsampling.cpp:5-97) with B = 1; with features/classes it also
 * replaces the numeric core of cpp_wrappers/cpp_subsampling (wrapper.cpp:58-286 ->
 * grid_subsampling/grid_subsampling.cpp:5-105).
 *
 *   points   f32[N,3]  stacked clouds          lens_dev  i32[B] points per batch element (device)
 *   features f32[N,fdim] or NULL (fdim 0)      classes   i32[N,ldim] or NULL (ldim 0)
 *   sub_points f32[N,3] (capacity N rows; the first M are valid)   sub_features f32[N,fdim]  sub_classes i32[N,ldim]
 *   sub_lens_dev i32[B]  voxels per element (device copy, feeds the next pyramid level)
 *   status_host i32[B+2] (HOST memory): [0] = M (total voxels), [1] = OR of D3F_ST_* flags, [2..] = voxels per element
 * Output is bit-identical to the reference INCLUDING row order (libstdc++ unordered_map iteration order).
 * The output size is data dependent, so this entry point synchronises `stream` ONCE internally (after the voxel
 * count is known) -- the reference op likewise allocates its output after computing it (tf_batch_subsampling.cpp:96-105).
 * When a flag is raised the outputs are not computed.
 * ------------------------------------------------------------------------------------------- */
size_t d3f_grid_subsample_workspace_bytes(int N, int B, int fdim, int ldim);
/* Capacity-mode form (points only): no host synchronisation.  `points` has N_cap rows of which sum(lens_dev) are valid;
 * sub_points has M_cap rows; elem_cap bounds the voxels of any ONE batch element (0: M_cap) -- the iteration-order rounds
 * are launched for that size, so a stack of B similar clouds should pass its per-cloud capacity; elem_points_cap bounds the
 * POINTS of any one batch element (0: N_cap): when it is at most 16384 every cloud is subsampled by ONE workgroup out of LDS
 * (two launches per call instead of ~25: the coarse pyramid levels); status_dev i32[2] (DEVICE) = [M, OR of D3F_ST_* flags];
 * M > M_cap, an element above elem_cap or above elem_points_cap raises D3F_ST_OUT_OVERFLOW (empty result).
 * Workspace: d3f_grid_subsample_workspace_bytes(N_cap, B, 0, 0). */
int d3f_batch_grid_subsample_async(const float* points, int N_cap, const int* lens_dev, int B, float dl,
                                   float* sub_points, int M_cap, int elem_cap, int elem_points_cap, int* sub_lens_dev,
                                   int* status_dev, void* workspace, size_t workspace_bytes, void* stream);
/* d3f_batch_grid_subsample_async with the clouds read IN PLACE: cloud b is its own device array clouds_dev[b] (f32[lens_dev[b], 3])
 * instead of rows of one stacked array -- the stage-0 call of a replayed fragment sequence takes its raw clouds where the
 * producer left them (no copy into a staging buffer: datasets/ThreeDMatch.py:186-192 hands the generator separate arrays too).
 * clouds_dev: device array of B device pointers; N_cap bounds sum(lens_dev).  Results are those of the stacked call. */
int d3f_batch_grid_subsample_async_inplace(const float* const* clouds_dev, int N_cap, const int* lens_dev, int B, float dl,
                                           float* sub_points, int M_cap, int elem_cap, int* sub_lens_dev, int* status_dev,
                                           void* workspace, size_t workspace_bytes, void* stream);

/* The self-pair stacking of the reference's test generators (datasets/ThreeDMatch.py:190-192, demo_registration.py:72-79:
 * np.concatenate([pts, pts])) for B stacked clouds whose row counts live in HBM (lens_in_dev i32[B]): cloud b is written
 * twice in a row -- out f32[2*M_cap,3] = [c_0; c_0; c_1; c_1; ...], lens_out_dev i32[2B] = [m_0, m_0, m_1, m_1, ...],
 * total_dev i32[1] = 2 * sum m_b.  B = 1: the single self-pair; B > 1: several fragments in one stack. */
int d3f_stack_self_pair(const float* pts, int M_cap, const int* lens_in_dev, int B, float* out, int* lens_out_dev,
                        int* total_dev, void* stream);
int d3f_batch_grid_subsample(const float* points, int N, const int* lens_dev, int B, float dl,
                             const float* features, int fdim, const int* classes, int ldim,
                             float* sub_points, float* sub_features, int* sub_classes, int* sub_lens_dev,
                             int* status_host, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Radius neighbours.
 * Replaces tf_custom_ops/tf_neighbors: BatchOrderedNeighbors (tf_batch_neighbors.cpp:8-30, kernel
 * :36-120 -> neighbors/neighbors.cpp:211-332 batch_nanoflann_neighbors / :125-208
 * batch_ordered_neighbors).  Per query: all supports of the same batch element with fp32
 * d2 = (dx*dx + dy*dy) + dz*dz < radius*radius (strict, no FMA), ordered by (d2, support index)
 * ascending -- exactly batch_ordered_neighbors, and equal to the active nanoflann path except inside
 * runs of bit-equal d2 (whose order nanoflann leaves unspecified).
 *
 *   queries f32[Nq,3], supports f32[Ns,3], q_lens_dev / s_lens_dev i32[B]
 *   out i32[Nq, ld]: columns [0,width) are written: the first min(count,width) neighbours then pad_value
 *       (pass Ns for BatchOrderedNeighbors, neighbors.cpp:324; -1 for OrderedNeighbors, neighbors.cpp:111-119)
 *   status_dev i32[2]: [0] = max neighbour count over all queries (the reference's output width Kmax),
 *                      [1] = OR of D3F_ST_* flags
 * `width` plays the role of datasets/common.py:399-406 (big_neighborhood_filter): pass the layer's
 * neighborhood limit; pass width = ld >= Kmax to get the untruncated matrix.
 * Nq / Ns are upper bounds (buffer rows): the real counts are sum(q_lens_dev) / sum(s_lens_dev), read on the device.
 * pad_value D3F_PAD_NUM_SUPPORTS pads with the device-side number of supports.
 * ------------------------------------------------------------------------------------------- */
size_t d3f_radius_neighbors_workspace_bytes(int Nq, int Ns, int B);
int d3f_batch_radius_neighbors(const float* queries, int Nq, const float* supports, int Ns,
                               const int* q_lens_dev, const int* s_lens_dev, int B, float radius,
                               int* out, int ld, int width, int pad_value, int* status_dev,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Two-step form of the same op.  datasets/common.py:1344,1367 search the SAME supports with the SAME radius twice
 * per pyramid level (conv_i and pool_i), so the cell grid is an object the caller keeps:
 *   d3f_neighbor_grid_build   supports -> cell grid (counting sort by cell) inside caller memory `grid`
 *                             (>= d3f_neighbor_grid_bytes(Ns, B) bytes, must stay untouched until the last search)
 *   d3f_neighbor_grid_search  queries against a built grid.
 *       queries_are_supports  1 when `queries` is the very array the grid was built from: queries are then visited in
 *                             cell order (L2 locality); results are identical either way
 *       cap                   in-radius supports per query that can be ordered (LDS budget; <= D3F_NEIGHBOR_CAP);
 *                             a query with more sets D3F_ST_HIT_OVERFLOW -> search again with a larger cap
 *       first_only            1: only column 0 (the nearest support, ties by index) is computed -- all that
 *                             closest_pool reads of the upsampling matrices (models/network_blocks.py:81);
 *                             columns 1..width-1 are filled with pad_value
 *       nn_hint               (first_only) > 0: the caller expects the nearest support within this distance (< radius), e.g.
 *                             sqrt(3) dl for the upsampling matrices, whose supports are the voxel barycentres of the queries:
 *                             only the cells that ball touches are visited first; a query whose nearest support is farther
 *                             is searched again over the full stencil, so the result never depends on the hint.  0: none
 *       reset_status          bit 0: status_dev is zeroed first (one extra launch); clear: the caller zeroed it (a
 *                             captured fragment zeroes the status words of all its ops with one fill).
 *                             bit 1 (D3F_NB_NO_KMAX): status_dev[0] (the largest neighbour count, which only callers that
 *                             size their output by it need) is NOT maintained -- every query otherwise reads one shared
 *                             word through L2, a same-address hot spot that serialises the whole launch
 */
size_t d3f_neighbor_grid_bytes(int Ns, int B);
/* byte offset inside a built grid of `order` i32[Ns]: the support indices sorted by cell.  Passing it as q_order /
 * row_order to the per-point kernels below makes them visit points in a spatially coherent order (neighbouring
 * workgroup rows share their gathered neighbours in L1 / L2); results do not depend on it. */
size_t d3f_neighbor_grid_order_offset(int Ns, int B);
int d3f_neighbor_grid_build(const float* supports, int Ns, const int* s_lens_dev, int B, float radius,
                            void* grid, size_t grid_bytes, void* stream);
int d3f_neighbor_grid_search(const void* grid, size_t grid_bytes, int Ns, const float* queries, int Nq,
                             const int* q_lens_dev, int B, float radius, int queries_are_supports,
                             int* out, int ld, int width, int pad_value, int cap, int first_only, float nn_hint,
                             int reset_status, int* status_dev, void* stream);
/* The general form of d3f_neighbor_grid_search: the queries may bring a grid of their own.
 *   query_grid   a grid built by d3f_neighbor_grid_build over `queries` themselves (any radius; Nq rows capacity, the same B), or
 *                `grid` itself when the queries ARE the supports, or NULL.  The queries are visited in ITS cell order: neighbouring
 *                wavefronts then read the same support runs.  Results do not depend on it -- unless D3F_NB_INTERNAL is set.
 *   flags        bit 0: zero status_dev first; bit 1 (D3F_NB_NO_KMAX); bit 2 (D3F_NB_INTERNAL, needs query_grid): the INTERNAL
 *                numbering of a pipeline that keeps every level in cell-sorted order -- row j of `out` belongs to the j-th query
 *                in the query grid's cell order (not to query j), and every entry is the POSITION of that support in `grid`'s cell
 *                order (d3f_neighbor_grid_inv_offset) instead of its index.  Which supports a row holds and their order (d2, then
 *                the ORIGINAL index) are exactly those of the reference numbering: renumbering both sides back gives the
 *                matrix of d3f_neighbor_grid_search bit for bit (tests/test_gpu_internal_order.py).  pad_value is written as is.
 *   first_only with D3F_NB_NO_KMAX: column 0 by the nearest-support kernel (four lanes per query). */
#define D3F_NB_INTERNAL 4
int d3f_neighbor_grid_search_ordered(const void* grid, size_t grid_bytes, int Ns, const float* queries, int Nq,
                                     const int* q_lens_dev, int B, float radius, const void* query_grid, size_t query_grid_bytes,
                                     int* out, int ld, int width, int pad_value, int cap, int first_only, float nn_hint, int flags,
                                     int* status_dev, void* stream);
/* byte offsets inside a built grid of `inv` i32[Ns] (position of support i in cell order: the inverse of `order`) and of `xyz`
 * f32[Ns, 3] (the supports in cell order: the point array of a level kept in the internal numbering). */
size_t d3f_neighbor_grid_inv_offset(int Ns, int B);
size_t d3f_neighbor_grid_xyz_offset(int Ns, int B);



/* ---------------------------------------------------------------------------------------------
 * KPConv, phase 1: neighbour gather + kernel-point influence + weighted aggregation.
 * Replaces the first half of kernels/convolution_ops.py:161-255 (KPConv_ops :186-240, :250-252):
 *   wf[n,p,c]  = sum_k  h(|| (s[idx[n,k]] - q[n]) - KP[p] ||) * f[idx[n,k], c]
 *   inv_cnt[n] = 1 / max(#{k : row_pos[idx[n,k]]}, 1),  row_pos from d3f_row_positive(f)
 * influence: 0 constant, 1 linear  h = max(1 - sqrt(d2+1e-10)/(2*KP_extent), 0), 2 gaussian (sigma = 0.3*KP_extent);
 * aggregation: 0 sum, 1 closest.  Shadow neighbours (idx >= Ns) contribute nothing.
 *   q f32[Nq,3]  s f32[Ns,3]  idx i32[Nq,ld_idx] (K columns used)  f f32[Ns,ldf] (Cin columns used)
 *   kp_host f32[num_kp,3] (HOST pointer: 45 floats passed by value to the kernel)  wf f32[Nq, num_kp*Cin]  inv_cnt f32[Nq]
 * Phase 2 is d3f_gemm_f32(wf, K_values reshaped [num_kp*Cin, Cout]) with row_scale = inv_cnt.
 * Nq_dev / Ns_dev (device i32, may be NULL): real row counts when Nq / Ns are capacities (see "Device-resident sizes").
 * ------------------------------------------------------------------------------------------- */
/* row_pos[s] = (sum_c f[s,c] > 0): the per-support test behind the neighbour count (:250-251), evaluated once
 * per support row.  row_pos u8[Ns]. */
/* feat_bf16 (here and in the KPConv / pooling / contraction entry points below): 0 = feature tensors are f32 (the parity path);
 * 1 = BASELINE configs[4] "bf16 features": every feature tensor named f / x / out / residual holds bfloat16 values (2 bytes per
 * element, leading dimensions in elements, 8-byte aligned rows); all arithmetic stays fp32, values are rounded to nearest even
 * when stored.  Only the shipped configuration (linear influence, 'sum', <= 15 kernel points) has the bf16 forms. */
int d3f_row_positive(const void* f, int Ns, int ldf, int Cin, unsigned char* row_pos, const int* Ns_dev, int feat_bf16,
                     void* stream);
int d3f_kpconv_aggregate(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                         const void* f, int ldf, int Cin, const unsigned char* row_pos, const float* kp_host,
                         int num_kp, float KP_extent, int influence, int aggregation, float* wf, float* inv_cnt,
                         const int* Nq_dev, const int* Ns_dev, const int* q_order, int feat_bf16, void* stream);

/* Whole KPConv_ops (kernels/convolution_ops.py:161-255) + the fused inference epilogue for Cin = 1 -- the input
 * layer of every shipped model (`simple` block on the all-ones features, models/network_blocks.py:222-244):
 *   out[n,o] = act( (sum_p wf[n,p] * W[p,o]) / max(#{k : f[idx[n,k]] > 0}, 1) * col_scale[o] + col_shift[o] + residual[n,o] )
 * f f32[Ns,1] (ldf), W f32[num_kp, Cout] (= K_values[:,0,:]); one kernel, only `out` is written to HBM. */
int d3f_kpconv_fused_c1(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                        const float* f, int ldf, const float* kp_host, int num_kp, float KP_extent, int influence,
                        int aggregation, const float* W, int Cout, const float* col_scale, const float* col_shift,
                        const float* residual, int ldr, int leaky, float alpha, void* out, int ldo,
                        const int* Nq_dev, const int* Ns_dev, const int* q_order, int out_bf16, void* stream);

/* Whole KPConv_ops + inference epilogue for Cin = Cout = 32 (the level-0 convolutions of the shipped architecture) in one
 * launch: gather + influences + aggregation as d3f_kpconv_aggregate, then the 32 x (num_kp*32) tile of weighted features is
 * contracted with K_values on the matrix cores straight from LDS -- the wf tensor (Nq x 480 floats) never reaches HBM.
 *   W f32[num_kp*32, 32] (= K_values reshaped, contiguous);  out f32[Nq, 32] (ldo);  row_pos from d3f_row_positive(f)
 *   out = act( (wf @ W) / max(count, 1) * col_scale + col_shift + residual ) */
int d3f_kpconv_fused32(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                       const void* f, int ldf, const unsigned char* row_pos, const float* kp_host, int num_kp,
                       float KP_extent, int influence, int aggregation, const float* W, const float* col_scale,
                       const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* out, int ldo,
                       const int* Nq_dev, const int* Ns_dev, const int* q_order, int feat_bf16, void* stream);

/* Whole KPConv_ops + epilogue in one kernel for Cin == Cout in {64, 128, 256} (levels 1 to 3 of the shipped architecture), the
 * [Nq, 15*Cin] weighted-feature tensor of kernels/convolution_ops.py:237-240 staying in LDS (tiles of 16 queries, passes of
 * 512 k-values, v_mfma_f32_16x16x4_f32).  Only the configuration of the shipped models (linear influence, 'sum'
 * aggregation, 15 kernel points): d3f_kpconv_fused_supported() says whether a call qualifies (1: use it; 2: the form
 * exists -- Cin = 256 -- but aggregation + contraction measured no slower; 0: not available); otherwise use
 * d3f_kpconv_aggregate + d3f_gemm_f32.  W_packed: K_values [15*Cin, Cout] reordered once by d3f_kpconv_pack_weights
 * (Wp[blk][g][n][j] = W[16*blk + 4*g + j][n]: the MFMA B-operand order, one 16-byte load per lane and k-block).
 * Arguments otherwise as d3f_kpconv_fused32.
 * Size limit of the one-kernel forms (d3f_kpconv_fused32 / d3f_kpconv_fused): rows are addressed with 24-bit multiplies, so
 * Nq, Ns, ld_idx, ldf < 2^24 and Nq * ld_idx, Ns * ldf < 2^31, else D3F_ERR_ARG (d3f_kpconv_aggregate + d3f_gemm_f32 have no
 * such limit: beyond it they take their generic kernel). */
int d3f_kpconv_fused_supported(int Cin, int Cout, int num_kp, int influence, int aggregation);
int d3f_kpconv_pack_weights(const float* W, int K, int N, float* W_packed, void* stream);
/* The same operator (kernels/convolution_ops.py:161-255 + epilogue) with its 15*Cin-deep contraction in the operand-split form of
 * d3f_gemm_x3 (round 5): the weighted features are written to LDS as three exact bfloat16 planes, K_values is pre-split once per
 * tensor by d3f_kpconv_pack_weights_x3 into d3f_kpconv_packed_x3_bytes(K, N) = 6 K N bytes (the B fragments of
 * v_mfma_f32_16x16x32_bf16; K % 32 == 0, N % 16 == 0), six exact products per fp32 product, fp32 accumulation: fp32 in, fp32 out,
 * 2.5 x less matrix-pipe time than v_mfma_f32_16x16x4_f32.  Arguments as d3f_kpconv_fused, W_packed = the planes. */
/* (N = 32: the fragment order of d3f_kpconv_fused32_x3, v_mfma_f32_32x32x16_bf16; any other N: d3f_kpconv_fused_x3's.) */
size_t d3f_kpconv_packed_x3_bytes(int K, int N);
int d3f_kpconv_pack_weights_x3(const float* W, int K, int N, void* W_planes, void* stream);
/* The same operator with the AGGREGATION on the matrix cores as well (round 6): wf[q] = influences^T x gathered features is a
 * [15 x K] x [K x 32] product per query -- v_mfma_f32_16x16x1_4b_f32, four queries per instruction, exact fp32 and in the reference's
 * neighbour order -- while the vector pipe computes the influences of the next neighbours.  The shipped configuration only
 * (num_kp = 15, linear influence, sum aggregation; fp32 features).  W_planes = d3f_kpconv_pack_weights_x3 of the [512, 32] matrix
 * W'[16 s + p][n] = K_values[p][c(s)][n] (p < 15) / 0 (p = 15), c(s) = 0, 2, ..., 30, 1, 3, ..., 31: the channel-major k order the
 * aggregation leaves on chip.
 * Other arguments as d3f_kpconv_fused32. */
int d3f_kpconv_fused32_mfma(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                            const float* f, int ldf, const unsigned char* rowpos, const float* kp_host, int num_kp,
                            float KP_extent, int influence, int aggregation, const void* W_planes, const float* col_scale,
                            const float* col_shift, const float* residual, int ldr, int leaky, float alpha, float* out,
                            int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order, void* stream);
/* d3f_kpconv_fused32 (Cin = Cout = 32, the level-0 convolutions) with the split contraction; W_planes =
 * d3f_kpconv_pack_weights_x3(K_values [15*32, 32]).  Arguments as d3f_kpconv_fused32. */
int d3f_kpconv_fused32_x3(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                          const void* f, int ldf, const unsigned char* rowpos, const float* kp_host, int num_kp,
                          float KP_extent, int influence, int aggregation, const float* W_planes, const float* col_scale,
                          const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* out,
                          int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order, int feat_bf16, void* stream);
int d3f_kpconv_fused_x3(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                        const void* f, int ldf, int Cin, const unsigned char* rowpos, const float* kp_host, int num_kp,
                        float KP_extent, int influence, int aggregation, const float* W_planes, int Cout,
                        const float* col_scale, const float* col_shift, const float* residual, int ldr, int leaky,
                        float alpha, void* out, int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order,
                        int feat_bf16, void* stream);
int d3f_kpconv_fused(const float* q, int Nq, const float* s, int Ns, const int* idx, int ld_idx, int K,
                     const void* f, int ldf, int Cin, const unsigned char* rowpos, const float* kp_host, int num_kp,
                     float KP_extent, int influence, int aggregation, const float* W_packed, int Cout,
                     const float* col_scale, const float* col_shift, const float* residual, int ldr, int leaky,
                     float alpha, void* out, int ldo, const int* Nq_dev, const int* Ns_dev, const int* q_order,
                     int feat_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense contraction on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, fmaf-chain numerics).
 * Replaces kernels/convolution_ops.py:90-99 (unary_convolution = tf.matmul) and :243-253 (the
 * kernel-weight contraction + neighbour-count normalisation), with the inference epilogue of
 * models/network_blocks.py:149-160,185-186 fused:
 *   C[m,n] = act( (sum_k A[m,k] B[k,n]) * row_scale[m] * col_scale[n] + col_shift[n] + residual[m,n] )
 * row_scale / col_scale / col_shift / residual may be NULL (identity); act = LeakyReLU(alpha) when
 * leaky != 0.  A f32[M,K] (lda), B f32[K,N] (ldb), C f32[M,N] (ldc), residual f32[M,N] (ldr).
 * workspace is used only when the call decides to split K (skinny shapes).
 * M_dev (device i32, may be NULL): real row count when M is a capacity; M_hint (0 = none): the row count the caller
 * expects, used only to plan the K split (pass the same value to d3f_gemm_workspace_bytes).
 * ------------------------------------------------------------------------------------------- */
size_t d3f_gemm_workspace_bytes(int M, int N, int K, int M_hint);
int d3f_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                 const float* row_scale, const float* col_scale, const float* col_shift,
                 const float* residual, int ldr, int leaky, float alpha,
                 void* workspace, size_t workspace_bytes, const int* M_dev, int M_hint, void* stream);

/* Decoder step: nearest upsampling + skip concatenation + the unary block that consumes them, as ONE contraction whose
 * A operand is composed on the fly (the concatenated tensor never exists in HBM).  Replaces models/D3Feat.py:39-63
 * (closest_pool, models/network_blocks.py:69-83, then tf.concat) followed by unary_block (models/network_blocks.py:207-219):
 *   C[m, :] = act( ([ x'[idx[m,0]] | skip[m] ] @ W) * col_scale + col_shift ),   x' = x with a zero row appended
 *   x f32[N1,C1] (ldx), idx i32[M, ld_idx] (column 0 is read), skip f32[M,C2] (lds; NULL when C2 = 0), W f32[C1+C2, N] (ldb)
 *   C1 % 4 == 0 when C2 > 0.  M_dev / N1_dev / M_hint as for d3f_gemm_f32; workspace: d3f_gemm_workspace_bytes(M, N, C1+C2, M_hint).
 * idx == NULL: no gather, A = [ x[m] | skip[m] ] -- used to contract the two branches of a resnet block in one launch
 * (models/network_blocks.py:321-368: conv3 + shortcut, batch-norm scales folded into the stacked weights). */
int d3f_gemm_upsample_cat_f32(const float* x, int N1, int ldx, int C1, const int* idx, int ld_idx,
                              const float* skip, int lds, int C2, const float* W, int ldb, float* C, int ldc,
                              int M, int N, const float* col_scale, const float* col_shift, int leaky, float alpha,
                              void* workspace, size_t workspace_bytes, const int* M_dev, const int* N1_dev,
                              int M_hint, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooling / upsampling gathers.
 *   d3f_ind_max_pool      models/network_blocks.py:51-66  out[n,c] = max_k x'[idx[n,k],c], x' = x + row of column minima
 *   d3f_closest_pool_cat  models/network_blocks.py:69-83 + models/D3Feat.py:63
 *                         out[n, 0:C1] = x'[idx[n,0]] (x' = x + zero row), out[n, C1:C1+C2] = skip[n]  (skip may be NULL, C2 0)
 *   col_min_dev: unused since round 6 (may be NULL): the column minima are taken inside the one pooling launch, by the
 *   threads of a row that has no valid neighbour and therefore takes the shadow row.
 * ------------------------------------------------------------------------------------------- */
int d3f_ind_max_pool(const void* x, int N1, int ldx, int C, const int* idx, int N2, int ld_idx, int K,
                     void* out, int ldo, float* col_min_dev, const int* N1_dev, const int* N2_dev, const int* row_order,
                     int feat_bf16, void* stream);
int d3f_closest_pool_cat(const float* x, int N1, int ldx, int C1, const int* idx, int N2, int ld_idx,
                         const float* skip, int lds, int C2, float* out, int ldo, const int* N1_dev, const int* N2_dev,
                         void* stream);

/* Stand-alone form of the GEMM epilogue (models/network_blocks.py:149-160 batch_norm in inference mode folded to
 * scale/shift, :185-186 leaky_relu, residual add):  out = act(x * col_scale + col_shift + residual). */
int d3f_affine_act(const float* x, int ldx, int M, int N, const float* col_scale, const float* col_shift,
                   const float* residual, int ldr, int leaky, float alpha, float* out, int ldo, const int* M_dev,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * D3Feat head: descriptors + detection scores.  Replaces models/D3Feat.py:65-115.
 *   x f32[N,C] (ldx)  un-normalised output of last_unary;  idx i32[N,ld_idx] level-0 neighbours (K columns)
 *   lens_dev i32[B] points per stacked cloud (B = 2 in the reference, any B >= 1 here)
 *   include_zero_dev i32[B] (device) or NULL: 1 if the cloud's row of in_batches contains the shadow index (so that
 *       the per-cloud maximum of :84-85 includes the zero row) -- datasets/common.py:453-496; NULL derives it from
 *       lens_dev on the device (shorter than the longest cloud, or all clouds equally long)
 *   stack_group: 0 = the B clouds are ONE reference stack (the rule above is applied over all of them); g > 0 = the
 *       stack is a concatenation of independent reference stacks of g consecutive clouds each (the fragment engine's
 *       batched replays: g = 2 pairs, g = 1 self-pairs computed once) and the rule is applied inside every group, so a
 *       fragment's result never depends on its stack mates.  Ignored when include_zero_dev is given.
 *   N is an upper bound: the real point count is sum(lens_dev)
 *   desc f32[N,C] = l2_normalize(x, eps 1e-10);  score f32[N]
 *   scratch_dev: >= 2*B + 2 + (N + 3) / 4 ints of device scratch (per-cloud maxima, offsets, one flag byte per row).  C <= 128.
 * ------------------------------------------------------------------------------------------- */
int d3f_detect_head(const float* x, int N, int ldx, int C, const int* idx, int ld_idx, int K,
                    const int* lens_dev, const int* include_zero_dev, int stack_group, int B, float* desc, int ldd,
                    float* score, int* scratch_dev, const int* row_order, void* stream);

/* The per-point output record of the path: out[n] = [ xyz(3) | desc(C) | score(1) ], f32, row stride ldo >= C + 4.
 * Replaces the three arrays the testers keep per fragment (utils/tester.py:215-229: points, features, scores written
 * side by side) with one contiguous block per fragment -- the unit the multi-GPU runner gathers once at the end.
 * N is an upper bound when N_dev (device int) is given. */
int d3f_pack_descriptors(const float* xyz, const float* desc, int ldd, int C, const float* score, int N, float* out,
                         int ldo, const int* N_dev, void* stream);
/* d3f_pack_descriptors with per-fragment destinations: the stack holds B clouds (lens_dev), a fragment is `group` consecutive
 * clouds; the rows of the first `keep` clouds of fragment f go to the address dst_ptrs_dev[f] (f32 rows of ldo floats, the
 * fragment's kept rows packed from row 0) when that entry is non-zero, every other row to `out` as in d3f_pack_descriptors.
 * What utils/tester.py:208-229 keeps of a stacked self-pair is its first cloud (in_batches[0]): group = 2, keep = 1 writes exactly
 * that, straight into the caller's shard buffer -- a replayed sequence needs no copy after it.
 * row_map_dev (optional, i32[N]): the inputs are in an internal row order (the cell order of the level-0 grid), record n belongs
 * at row row_map_dev[n] of the reference order -- the one place where a pipeline on the internal numbering returns to the
 * reference's row order.  dst_ptrs_dev may be NULL when only the row map is wanted. */
int d3f_pack_descriptors_to(const float* xyz, const float* desc, int ldd, int C, const float* score, int N, float* out,
                            int ldo, const int* N_dev, const int* lens_dev, int B, int group, int keep,
                            const long long* dst_ptrs_dev, const int* row_map_dev, void* stream);


/* LDS-DMA form of the fp32 contractions (round 4): the operator of d3f_gemm_f32 / d3f_gemm_upsample_cat_f32 with the same
 * composite A operand as d3f_gemm_bf16 below -- A f32[., C1] (rows in place when idx == NULL, else the gathered rows x'[idx[m,0]],
 * zero row for indices outside [0, N1)), optional second operand skip f32[M, C2], K = C1 + C2 -- and the same epilogue
 * (kernels/convolution_ops.py:90-99,243-253 + models/network_blocks.py:149-160,185-186 + the resnet residual add), exact fp32
 * products and sums on v_mfma_f32_32x32x2_f32.  The weights are handed over PRE-TRANSPOSED: Wt = d3f_gemm_pack_f32t(W f32[K,N])
 * -> f32 [N][Kp], Kp = K rounded up to 32, zero padded (4 * N * Kp bytes), made once per weight tensor.  Requires float4-addressable
 * operands (C1, C2, N, lda, lds, ldc, ldr multiples of 4; 16-byte aligned bases): D3F_ERR_ARG otherwise -- use d3f_gemm_f32 then.
 * workspace >= d3f_gemm_workspace_bytes(M, N, K, M_hint). */
int d3f_gemm_pack_f32t(const float* W, int ldb, int K, int N, float* Wt, void* stream);
int d3f_gemm_f32t(const float* A, int N1, int lda, int C1, const int* idx, int ld_idx, const float* skip, int lds, int C2,
                  const float* Wt, float* C, int ldc, int M, int N, const float* row_scale, const float* col_scale,
                  const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* workspace,
                  size_t workspace_bytes, const int* M_dev, const int* N1_dev, int M_hint, void* stream);

/* Operand-split form of the fp32 contractions: the operator, operands and epilogue of d3f_gemm_f32t, with every fp32 operand value
 * written EXACTLY as the sum of three bfloat16 values (8 + 8 + 8 significant bits) and every fp32 product as the six exact bf16
 * products whose sum differs from it by < 2^-23 relative, accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (d3feat_amd/csrc/gemm_x3.h).
 * fp32 in, fp32 out, fp32-grade error (tests/test_gpu_gemm_x3.py measures it against float64 beside d3f_gemm_f32t's); the weights are
 * handed over PRE-SPLIT: Wx = d3f_gemm_pack_x3(W f32[K,N]), d3f_gemm_x3_packed_bytes(K, N) bytes, made once per weight tensor.
 * On top of d3f_gemm_f32t's addressing rules: K = C1 + C2 a multiple of 32 and, with a second operand, C1 a multiple of 32 too --
 * D3F_ERR_ARG otherwise (use d3f_gemm_f32t).  workspace >= d3f_gemm_x3_workspace_bytes(M, N, K, M_hint). */
size_t d3f_gemm_x3_packed_bytes(int K, int N);
int d3f_gemm_pack_x3(const float* W, int ldb, int K, int N, void* Wx, void* stream);
size_t d3f_gemm_x3_workspace_bytes(int M, int N, int K, int M_hint);
/* the workgroup shape (rows x cols: 128 x 32, 128 x 64 or 256 x 128) and K slice count d3f_gemm_x3 uses for a problem; D3F_ERR_ARG
 * for a K it does not take.  Diagnostics / tests only: the launcher decides by itself. */
int d3f_gemm_x3_plan(int M, int N, int K, int M_hint, int* rows, int* cols, int* slices);
int d3f_gemm_x3(const float* A, int N1, int lda, int C1, const int* idx, int ld_idx, const float* skip, int lds, int C2,
                const void* Wx, float* C, int ldc, int M, int N, const float* row_scale, const float* col_scale,
                const float* col_shift, const float* residual, int ldr, int leaky, float alpha, void* workspace,
                size_t workspace_bytes, const int* M_dev, const int* N1_dev, int M_hint, void* stream);

/* bf16-operand form of the contractions (BASELINE.json configs[4]: batched inference, bf16 MFMA contraction): the operator of
 * d3f_gemm_f32 / d3f_gemm_upsample_cat_f32 -- A f32[M, C1] (rows in place when idx == NULL, else the gathered rows
 * x'[idx[m,0]], zero row for indices outside [0, N1)), optional second operand skip f32[M, C2], K = C1 + C2, same epilogue --
 * with both operands rounded to bfloat16 (nearest even) and multiplied by v_mfma_f32_32x32x16_bf16, fp32 accumulate.
 * NOT bit-compatible with the fp32 path (2^-9 relative operand rounding); separate tolerance, separate bench configuration.
 * W_packed: d3f_gemm_pack_bf16(W f32[K,N]) -> bf16 [N][Kp], Kp = K rounded up to 32 (2 * N * Kp bytes).
 * C1, C2, lda, lds multiples of 4, 16-byte aligned bases.  workspace >= d3f_gemm_bf16_workspace_bytes(M, N, K, M_hint) -- its
 * OWN sizing function: the bf16 kernel's tile and K split differ from the fp32 kernel's (round-3 header pointed at
 * d3f_gemm_workspace_bytes, which under-sizes the slabs for N <= 32). */
size_t d3f_gemm_bf16_workspace_bytes(int M, int N, int K, int M_hint);
int d3f_gemm_pack_bf16(const float* W, int ldb, int K, int N, void* W_packed, void* stream);
/* a_bf16: A, skip and residual hold bfloat16 values (bf16 feature storage, see d3f_row_positive); c_bf16: C is written as bfloat16.
 * Both 0: the operands are f32 in HBM and only rounded on their way into the multiply (the round-2 form of configs[4]). */
int d3f_gemm_bf16(const void* A, int N1, int lda, int C1, const int* idx, int ld_idx, const void* skip, int lds, int C2,
                  const void* W_packed, void* C, int ldc, int M, int N, const float* row_scale, const float* col_scale,
                  const float* col_shift, const void* residual, int ldr, int leaky, float alpha, void* workspace,
                  size_t workspace_bytes, const int* M_dev, const int* N1_dev, int M_hint, int a_bf16, int c_bf16, void* stream);

/* Stage-0 ingestion: float32 xyz [n,3] out of raw file records already on the device -- a binary PLY
 * vertex element (demo_registration.py:23, datasets/ThreeDMatch.py:348; float or double coordinates at byte offsets
 * off_x/y/z of `stride`-byte records, either byte order) or a KITTI velodyne sweep (datasets/KITTI.py:131, 277-278:
 * stride 16, offsets 0/4/8).  The output feeds d3f_batch_grid_subsample directly. */
int d3f_decode_xyz_records(const void* raw, int n, int stride, int off_x, int off_y, int off_z, int is_f64, int big_endian,
                           float* out, void* stream);

/* =============================================================================================
 * Downstream matching -- what the reference does with the descriptors after the hot path.
 * ============================================================================================= */

/* Nearest descriptor: idx[i] = argmin_j ||A_i - B_j||^2 (lowest j on ties, -1 when Nb == 0); d2_out (optional) the minimum.
 * Replaces the argmin over the dense distance matrix of geometric_registration/evaluate.py:17-21 (one direction per call)
 * and the KD-tree feature lookup inside open3d.registration_ransac_based_on_feature_matching (evaluate.py:93-99).
 * A f32[Na,C] (lda), B f32[Nb,C] (ldb), C in {16, 32, 64}. */
size_t d3f_feature_nn_workspace_bytes(int Na);
int d3f_feature_nn(const float* A, int Na, int lda, const float* B, int Nb, int ldb, int C, int* idx, float* d2_out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Mutually closest pairs (evaluate.py:21-26 build_correspondence): pairs (i, ab[i]) with ba[ab[i]] == i, ascending i.
 * pairs i32[<= Na, 2]; count_dev i32[1] on the device. */
size_t d3f_mutual_matches_workspace_bytes(int Na);
int d3f_mutual_matches(const int* ab, int Na, const int* ba, int Nb, int* pairs, int* count_dev, void* workspace,
                       size_t workspace_bytes, void* stream);

/* RANSAC hypotheses of open3d.registration_ransac_based_on_feature_matching (demo_registration.py:184-192, evaluate.py:
 * 93-99) for iterations it0 .. it0+H-1: sample ransac_n source points (counter-based random numbers: a pure function of
 * (seed, iteration, draw) -- d3f_ransac_draw is the host copy), pair each with nn[.] (its nearest target FEATURE),
 * CorrespondenceCheckerBasedOnEdgeLength(edge_similarity; <= 0: off), rigid fit without scaling
 * (TransformationEstimationPointToPoint(False)), CorrespondenceCheckerBasedOnDistance(checker_distance; <= 0: off).
 * T_out f32[H,12] row-major [R | t]; valid_out u8[H]. */
int d3f_ransac_hypotheses(const float* src, int Ns, const float* tgt, int Nt, const int* nn, int ransac_n,
                          float edge_similarity, float checker_distance, uint64_t seed, uint64_t it0,
                          int H, float* T_out, unsigned char* valid_out, void* stream);
int d3f_ransac_draw(uint64_t seed, uint64_t iteration, int d, int n);

/* Fitness of V hypotheses (Open3D's GetRegistrationResultAndCorrespondences): for every transformed source point the
 * nearest TARGET point strictly inside `radius` (grid built over the target by d3f_neighbor_grid_build, one cloud);
 * count_dev i32[V] inliers, sumd2_dev u64[V] sum of squared distances in 2^-32 units (order independent),
 * nearest_dev i32[Ns] (optional) the correspondences under hypothesis 0. */
int d3f_neighbor_grid_score(const void* grid, size_t grid_bytes, int Nt, const float* src, int Ns, const float* T, int V,
                            float radius, int* count_dev, uint64_t* sumd2_dev, int* nearest_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3FEAT_AMD_H */
