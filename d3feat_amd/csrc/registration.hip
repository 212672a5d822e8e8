// Downstream matching on gfx950: what the reference does with the descriptors after the hot path.
//
//   * feature nearest neighbours / mutual matches   geometric_registration/evaluate.py:11-27 (build_correspondence:
//       argmin over a [n x m] distance matrix of 32-d descriptors, both directions, keep the mutually closest pairs);
//   * RANSAC on feature matches                      evaluate.py:93-99, demo_registration.py:184-192
//       (open3d.registration_ransac_based_on_feature_matching: sample ransac_n source points, pair each with its nearest
//       target FEATURE, edge-length checker, rigid fit (no scaling), distance checker, score the fit by the nearest target
//       POINT of every transformed source point within max_correspondence_distance; best fitness, then lowest rmse).
//
// Open3D 0.7 is third-party code outside /root/reference with unspecified random sampling, so results are pinned to this
// repo's numpy restatement of the SAME algorithm with the SAME counter-based random numbers (oracle/registration_np.py),
// not to Open3D's stream: hypotheses are a pure function of (seed, iteration).
//
// Kernels: distance tiles on the VALU (C = 32: 64 flops per pair, B tile broadcast from LDS, row minima merged across column
// splits with one 64-bit atomicMin per row -- key = d2 bits << 32 | column, ties to the lowest column); one thread per RANSAC
// hypothesis (Horn's closed-form absolute orientation: largest eigenvector of a 4x4 symmetric matrix by cyclic Jacobi, fp64);
// scoring of the validated hypotheses against the neighbour grid of radius_neighbors.hip (d3f_neighbor_grid_score).
#include "prims.h"

// ---------------------------------------------------------------------------------------------------------------------
// nearest feature: key[i] = min_j (||A_i - B_j||^2 bits << 32 | j)
// ---------------------------------------------------------------------------------------------------------------------
#define RG_TB 128   // B rows per LDS tile

template <int C>
__global__ void __launch_bounds__(256) rg_feature_nn_kernel(const float* __restrict__ A, int Na, int lda,
                                                            const float* __restrict__ B, int Nb, int ldb,
                                                            unsigned long long* __restrict__ key) {
    __shared__ float tile[RG_TB * C];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a[C];
#pragma unroll
    for (int c = 0; c < C; ++c) a[c] = (i < Na) ? A[(size_t)i * lda + c] : 0.f;
    const int per = (Nb + gridDim.y - 1) / gridDim.y;
    const int j0 = blockIdx.y * per, j1 = min(Nb, j0 + per);
    float best = 3.402823466e38f;
    int bj = 0x7fffffff;
    for (int t0 = j0; t0 < j1; t0 += RG_TB) {
        const int nt = min(RG_TB, j1 - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < nt * C; e += 256) tile[e] = B[(size_t)(t0 + e / C) * ldb + (e % C)];
        __syncthreads();
        for (int j = 0; j < nt; ++j) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = a[c] - tile[j * C + c];
                d2 = fmaf(d, d, d2);
            }
            if (d2 < best) { best = d2; bj = t0 + j; }   // strict: ties keep the lowest column
        }
    }
    if (i < Na && bj != 0x7fffffff)
        atomicMin(&key[i], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bj);
}

__global__ void __launch_bounds__(256) rg_unpack_kernel(const unsigned long long* __restrict__ key, int N, int* __restrict__ idx,
                                                        float* __restrict__ d2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const unsigned long long k = key[i];
    idx[i] = (k == ~0ull) ? -1 : (int)(k & 0xffffffffull);
    if (d2) d2[i] = (k == ~0ull) ? 3.402823466e38f : __uint_as_float((unsigned)(k >> 32));
}

extern "C" size_t d3f_feature_nn_workspace_bytes(int Na) { return d3f_align((size_t)(Na > 0 ? Na : 1) * 8) + 256; }

// idx[i] = argmin_j ||A_i - B_j||^2 (lowest j on ties; -1 when Nb == 0), d2_out[i] (optional) the minimum.  C in {16, 32, 64}.
extern "C" int d3f_feature_nn(const float* A, int Na, int lda, const float* B, int Nb, int ldb, int C, int* idx, float* d2_out,
                              void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Na < 0 || Nb < 0 || lda < C || ldb < C || (C != 16 && C != 32 && C != 64)) return D3F_ERR_ARG;
    if (Na == 0) return D3F_OK;
    if (!A || !idx || (Nb > 0 && !B)) return D3F_ERR_ARG;
    if (!workspace || workspace_bytes < (size_t)Na * 8) return D3F_ERR_WORKSPACE;
    unsigned long long* key = (unsigned long long*)workspace;
    int rc = d3f_fill_u32(key, (size_t)Na * 2, 0xFFFFFFFFu, stream);
    if (rc != D3F_OK) return rc;
    if (Nb > 0) {
        const int bx = d3f_cdiv(Na, 256);
        int by = d3f_cdiv(1024, bx);                       // ~1024 workgroups: four per CU
        const int by_max = d3f_cdiv(Nb, RG_TB);
        if (by > by_max) by = by_max;
        if (by < 1) by = 1;
        dim3 grid(bx, by);
        if (C == 16) rg_feature_nn_kernel<16><<<grid, 256, 0, stream>>>(A, Na, lda, B, Nb, ldb, key);
        else if (C == 32) rg_feature_nn_kernel<32><<<grid, 256, 0, stream>>>(A, Na, lda, B, Nb, ldb, key);
        else rg_feature_nn_kernel<64><<<grid, 256, 0, stream>>>(A, Na, lda, B, Nb, ldb, key);
    }
    rg_unpack_kernel<<<d3f_cdiv(Na, 256), 256, 0, stream>>>(key, Na, idx, d2_out);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// mutual matches (evaluate.py:21-26): pairs (i, ab[i]) with ba[ab[i]] == i, in ascending i
// ---------------------------------------------------------------------------------------------------------------------
struct RgMutualIn {
    const int* ab; const int* ba; int Nb;
    __device__ __forceinline__ int operator()(int i) const {
        const int j = ab[i];
        return (j >= 0 && j < Nb && ba[j] == i) ? 1 : 0;
    }
};
struct RgCountEpi {
    int* count;
    __device__ __forceinline__ void operator()(int total) const { if (threadIdx.x == 0) *count = total; }
};
__global__ void __launch_bounds__(256) rg_mutual_write_kernel(RgMutualIn in, int Na, const int* __restrict__ local,
                                                              const int* __restrict__ base, int* __restrict__ pairs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Na || !in(i)) return;
    const int p = d3f_scan_at(local, base, i);
    pairs[2 * p] = i;
    pairs[2 * p + 1] = in.ab[i];
}

extern "C" size_t d3f_mutual_matches_workspace_bytes(int Na) {
    return d3f_align((size_t)(Na > 0 ? Na : 1) * 4) + d3f_align(d3f_scan_base_ints(Na) * 4) + 512;
}

// pairs i32[<= Na, 2], count_dev i32[1] (device).  ab i32[Na] (A -> B nearest), ba i32[Nb] (B -> A nearest).
extern "C" int d3f_mutual_matches(const int* ab, int Na, const int* ba, int Nb, int* pairs, int* count_dev, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Na < 0 || Nb < 0 || !count_dev) return D3F_ERR_ARG;
    if (Na == 0) return d3f_fill_u32(count_dev, 1, 0u, stream);
    if (!ab || !ba || !pairs) return D3F_ERR_ARG;
    D3fArena ar(workspace, workspace_bytes);
    int* local = ar.take<int>(Na);
    int* base = ar.take<int>(d3f_scan_base_ints(Na));
    unsigned* counter = ar.take<unsigned>(4);
    if (!ar.ok) return D3F_ERR_WORKSPACE;
    int rc = d3f_fill_u32(counter, 4, 0u, stream);
    if (rc != D3F_OK) return rc;
    RgMutualIn in{ab, ba, Nb};
    if ((rc = d3f_scan_fold_launch(in, Na, nullptr, local, base, counter, RgCountEpi{count_dev}, stream)) != D3F_OK) return rc;
    rg_mutual_write_kernel<<<d3f_cdiv(Na, 256), 256, 0, stream>>>(in, Na, local, base, pairs);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// RANSAC hypotheses: one thread per iteration
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long rg_splitmix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// draw d of iteration it: uniform integer in [0, n)  (counter based: a pure function of (seed, it, d))
__host__ __device__ __forceinline__ int rg_draw(unsigned long long seed, unsigned long long it, int d, int n) {
    const unsigned long long r = rg_splitmix(seed ^ rg_splitmix(it * 64ull + (unsigned long long)d));
    return (int)((r >> 11) % (unsigned long long)n);
}

// Horn 1987: rotation maximising sum t_i . R s_i from the cross-covariance S = sum (s_i - ms)(t_i - mt)^T: unit quaternion =
// eigenvector of the largest eigenvalue of the symmetric 4x4 matrix N(S); cyclic Jacobi in fp64 (a fixed number of sweeps).
__device__ void rg_horn(const double S[3][3], double R[3][3]) {
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {0, S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {0, 0, -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {0, 0, 0, -S[0][0] - S[1][1] + S[2][2]}};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < i; ++j) N[i][j] = N[j][i];
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = N[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (N[q][q] - N[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {   // columns p, q of N
                    const double nkp = N[k][p], nkq = N[k][q];
                    N[k][p] = c * nkp - s * nkq;
                    N[k][q] = s * nkp + c * nkq;
                }
                for (int k = 0; k < 4; ++k) {   // rows p, q of N
                    const double npk = N[p][k], nqk = N[q][k];
                    N[p][k] = c * npk - s * nqk;
                    N[q][k] = s * npk + c * nqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int i = 1; i < 4; ++i) if (N[i][i] > N[m][m]) m = i;
    double w = V[0][m], x = V[1][m], y = V[2][m], z = V[3][m];
    const double nrm = sqrt(w * w + x * x + y * y + z * z);
    w /= nrm; x /= nrm; y /= nrm; z /= nrm;
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z);     R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);     R[2][1] = 2 * (y * z + w * x);     R[2][2] = 1 - 2 * (x * x + y * y);
}

#define RG_MAXN 8

// T[h] = 12 floats (row-major 3x4 [R | t]); valid[h] = 1 iff the samples are distinct, every sample has a match, the
// edge-length checker (if edge_sim > 0) and, after the fit, the distance checker (if dist_thr > 0) pass.
__global__ void __launch_bounds__(256) rg_hypotheses_kernel(const float* __restrict__ src, int Ns, const float* __restrict__ tgt,
                                                            int Nt, const int* __restrict__ nn, int n,
                                                            float edge_sim, float dist_thr, unsigned long long seed,
                                                            unsigned long long it0, int H, float* __restrict__ T,
                                                            unsigned char* __restrict__ valid) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const unsigned long long it = it0 + (unsigned long long)h;
    int si[RG_MAXN], ti[RG_MAXN];
    bool ok = true;
    for (int d = 0; d < n; ++d) {
        si[d] = rg_draw(seed, it, d, Ns);
        ti[d] = nn[si[d]];
        if (ti[d] < 0 || ti[d] >= Nt) ok = false;
        for (int e = 0; e < d; ++e) if (si[e] == si[d]) ok = false;
    }
    double s[RG_MAXN][3], t[RG_MAXN][3];
    if (ok) {
        for (int d = 0; d < n; ++d)
            for (int c = 0; c < 3; ++c) { s[d][c] = (double)src[3 * (size_t)si[d] + c]; t[d][c] = (double)tgt[3 * (size_t)ti[d] + c]; }
        if (edge_sim > 0.f) {   // CorrespondenceCheckerBasedOnEdgeLength: every pair of edges similar in length both ways
            for (int a = 0; a < n && ok; ++a)
                for (int b = a + 1; b < n; ++b) {
                    double ds = 0, dt = 0;
                    for (int c = 0; c < 3; ++c) { ds += (s[a][c] - s[b][c]) * (s[a][c] - s[b][c]); dt += (t[a][c] - t[b][c]) * (t[a][c] - t[b][c]); }
                    ds = sqrt(ds); dt = sqrt(dt);
                    if (ds < dt * (double)edge_sim || dt < ds * (double)edge_sim) { ok = false; break; }
                }
        }
    }
    float out[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (ok) {
        double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
        for (int d = 0; d < n; ++d) for (int c = 0; c < 3; ++c) { ms[c] += s[d][c]; mt[c] += t[d][c]; }
        for (int c = 0; c < 3; ++c) { ms[c] /= n; mt[c] /= n; }
        double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int d = 0; d < n; ++d)
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] += (s[d][a] - ms[a]) * (t[d][b] - mt[b]);
        double R[3][3];
        rg_horn(S, R);
        double tr[3];
        for (int a = 0; a < 3; ++a) tr[a] = mt[a] - (R[a][0] * ms[0] + R[a][1] * ms[1] + R[a][2] * ms[2]);
        if (dist_thr > 0.f) {   // CorrespondenceCheckerBasedOnDistance on the aligned samples
            for (int d = 0; d < n; ++d) {
                double e2 = 0;
                for (int a = 0; a < 3; ++a) {
                    const double v = R[a][0] * s[d][0] + R[a][1] * s[d][1] + R[a][2] * s[d][2] + tr[a] - t[d][a];
                    e2 += v * v;
                }
                if (sqrt(e2) > (double)dist_thr) ok = false;
            }
        }
        for (int a = 0; a < 3; ++a) { out[4 * a] = (float)R[a][0]; out[4 * a + 1] = (float)R[a][1]; out[4 * a + 2] = (float)R[a][2]; out[4 * a + 3] = (float)tr[a]; }
    }
    for (int k = 0; k < 12; ++k) T[(size_t)h * 12 + k] = out[k];
    valid[h] = ok ? 1 : 0;
}

// H hypotheses for iterations it0 .. it0 + H - 1.  nn i32[Ns]: nearest target FEATURE of every source point.
extern "C" int d3f_ransac_hypotheses(const float* src, int Ns, const float* tgt, int Nt, const int* nn, int ransac_n,
                                     float edge_similarity, float checker_distance, uint64_t seed, uint64_t it0, int H,
                                     float* T_out, unsigned char* valid_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Ns < 1 || Nt < 1 || ransac_n < 3 || ransac_n > RG_MAXN || H < 0) return D3F_ERR_ARG;
    if (H == 0) return D3F_OK;
    if (!src || !tgt || !nn || !T_out || !valid_out) return D3F_ERR_ARG;
    rg_hypotheses_kernel<<<d3f_cdiv(H, 256), 256, 0, stream>>>(src, Ns, tgt, Nt, nn, ransac_n, edge_similarity, checker_distance,
                                                               seed, it0, H, T_out, valid_out);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
}

// host copy of the sampler (bindings / tests draw the very same indices)
extern "C" int d3f_ransac_draw(uint64_t seed, uint64_t iteration, int d, int n) {
    return n > 0 ? rg_draw(seed, iteration, d, n) : -1;
}
