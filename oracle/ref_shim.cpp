// TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path.
//
// extern "C" shim over the reference's own C++ cores, compiled IN PLACE from
// /root/reference (nothing is copied into this repo):
//   tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp        (batch_nanoflann_neighbors :211-332,
//                                                              batch_ordered_neighbors :125-208,
//                                                              ordered_neighbors :58-123)
//   tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp  (:5-97, :101-149)
//   tf_custom_ops/cpp_utils/cloud/cloud.cpp
// The argument marshalling mirrors what the TF op wrappers do
// (tf_batch_neighbors.cpp:75-115, tf_batch_subsampling.cpp:56-120): copy into std::vector, call, copy out.
// Built by oracle/Makefile into oracle/_ref/libd3f_ref.so (git-ignored).
#include "tf_custom_ops/tf_neighbors/neighbors/neighbors.h"
#include "tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.h"
#include <cstring>
#include <cstdlib>

static std::vector<PointXYZ> to_pts(const float* p, int n) {
    return std::vector<PointXYZ>((const PointXYZ*)p, (const PointXYZ*)p + n);
}

extern "C" {

// Returns Kmax; *out is malloc'ed int[Nq*Kmax] (free with ref_free).
int ref_batch_nanoflann_neighbors(const float* q, int Nq, const float* s, int Ns,
                                  const int* qb, const int* sb, int B, float radius, int** out) {
    std::vector<PointXYZ> Q = to_pts(q, Nq), S = to_pts(s, Ns);
    std::vector<int> QB(qb, qb + B), SB(sb, sb + B), res;
    batch_nanoflann_neighbors(Q, S, QB, SB, res, radius);
    int k = Nq ? (int)(res.size() / Nq) : 0;
    *out = (int*)malloc(sizeof(int) * (res.size() + 1));
    memcpy(*out, res.data(), sizeof(int) * res.size());
    return k;
}

int ref_batch_ordered_neighbors(const float* q, int Nq, const float* s, int Ns,
                                const int* qb, const int* sb, int B, float radius, int** out) {
    std::vector<PointXYZ> Q = to_pts(q, Nq), S = to_pts(s, Ns);
    std::vector<int> QB(qb, qb + B), SB(sb, sb + B), res;
    batch_ordered_neighbors(Q, S, QB, SB, res, radius);
    int k = Nq ? (int)(res.size() / Nq) : 0;
    *out = (int*)malloc(sizeof(int) * (res.size() + 1));
    memcpy(*out, res.data(), sizeof(int) * res.size());
    return k;
}

int ref_ordered_neighbors(const float* q, int Nq, const float* s, int Ns, float radius, int** out) {
    std::vector<PointXYZ> Q = to_pts(q, Nq), S = to_pts(s, Ns);
    std::vector<int> res;
    ordered_neighbors(Q, S, res, radius);
    int k = Nq ? (int)(res.size() / Nq) : 0;
    *out = (int*)malloc(sizeof(int) * (res.size() + 1));
    memcpy(*out, res.data(), sizeof(int) * res.size());
    return k;
}

// Returns M; *out is malloc'ed float[M*3].
int ref_grid_subsampling(const float* p, int N, float dl, float** out) {
    std::vector<PointXYZ> P = to_pts(p, N), R;
    std::vector<float> f, rf;
    std::vector<int> c, rc;
    grid_subsampling(P, R, f, rf, c, rc, dl);
    *out = (float*)malloc(sizeof(float) * 3 * (R.size() + 1));
    memcpy(*out, R.data(), sizeof(float) * 3 * R.size());
    return (int)R.size();
}

// Returns M; *out float[M*3]; out_b int[B].
int ref_batch_grid_subsampling(const float* p, int N, const int* b, int B, float dl, float** out, int* out_b) {
    std::vector<PointXYZ> P = to_pts(p, N), R;
    std::vector<float> f, rf;
    std::vector<int> c, rc, OB(b, b + B), RB;
    batch_grid_subsampling(P, R, f, rf, c, rc, OB, RB, dl);
    *out = (float*)malloc(sizeof(float) * 3 * (R.size() + 1));
    memcpy(*out, R.data(), sizeof(float) * 3 * R.size());
    for (int i = 0; i < B; i++) out_b[i] = RB[i];
    return (int)R.size();
}

void ref_free(void* p) { free(p); }

}  // extern "C"
