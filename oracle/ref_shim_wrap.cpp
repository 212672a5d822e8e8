// TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path.
//
// extern "C" shim over the CPython extension's core
//   cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-105   (8-argument form, with `verbose`)
// compiled in place from /root/reference.  wrapper.cpp itself does not build against numpy 2.x
// (SURVEY.md §8c), so the argument handling of wrapper.cpp:70-276 is restated in the Python
// mirror and only the numeric core is taken from the reference.  Separate .so from libd3f_ref.so
// because both cores define a different `class SampledData`.
#include "cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include <cstring>
#include <cstdlib>

extern "C" {

// features: float[N*fdim] or NULL (fdim 0); classes: int[N*ldim] or NULL (ldim 0).
// Returns M; outputs malloc'ed (free with refw_free).
int refw_grid_subsampling(const float* p, int N, const float* feat, int fdim, const int* cls, int ldim,
                          float dl, float** out_p, float** out_f, int** out_c) {
    std::vector<PointXYZ> P((const PointXYZ*)p, (const PointXYZ*)p + N), R;
    std::vector<float> f, rf;
    std::vector<int> c, rc;
    if (fdim > 0) f.assign(feat, feat + (size_t)N * fdim);
    if (ldim > 0) c.assign(cls, cls + (size_t)N * ldim);
    grid_subsampling(P, R, f, rf, c, rc, dl, 0);
    size_t M = R.size();
    *out_p = (float*)malloc(sizeof(float) * 3 * (M + 1));
    memcpy(*out_p, R.data(), sizeof(float) * 3 * M);
    *out_f = (float*)malloc(sizeof(float) * (rf.size() + 1));
    memcpy(*out_f, rf.data(), sizeof(float) * rf.size());
    *out_c = (int*)malloc(sizeof(int) * (rc.size() + 1));
    memcpy(*out_c, rc.data(), sizeof(int) * rc.size());
    return (int)M;
}

void refw_free(void* p) { free(p); }

}
