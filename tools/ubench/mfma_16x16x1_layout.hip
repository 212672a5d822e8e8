// Register layout of v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 outer products per instruction), found empirically:
// block b of A / B = lanes 16 b .. 16 b + 15; prints, for every (lane, register) of D, which (block, i, j) it holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A[b][i] = 1000 (b+1) + i ; B[b][j] = 1 + j / 64.0  -> product identifies b, i, j
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    // probe i: A = one-hot over (b, i) encodes; run 3 MFMAs with different encodings
    const int b = l / 16, i = l % 16;
    f16v cb = c, ci = c, cj = c;
    cb = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(b + 1), 1.0f, cb, 0, 0, 0);      // D = block of A
    ci = __builtin_amdgcn_mfma_f32_16x16x1f32((float)i, 1.0f, ci, 0, 0, 0);            // D = row i
    cj = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)i, cj, 0, 0, 0);            // D = column j (B's lane index)
    f16v cbb = c;
    cbb = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)(b + 1), cbb, 0, 0, 0);    // D = block of B
    for (int r = 0; r < 16; ++r) {
        out[(l * 16 + r) * 4 + 0] = cb[r]; out[(l * 16 + r) * 4 + 1] = ci[r]; out[(l * 16 + r) * 4 + 2] = cj[r];
        out[(l * 16 + r) * 4 + 3] = cbb[r];
    }
}
int main() {
    float* d; float h[64 * 16 * 4];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const float* e = &h[(l * 16 + r) * 4];
            const int blkA = (int)e[0] - 1, i = (int)e[1], j = (int)e[2], blkB = (int)e[3] - 1;
            const int want_b = r / 4, want_i = 4 * (l / 16) + r % 4, want_j = l % 16;
            if (blkA != want_b || blkB != want_b || i != want_i || j != want_j) ok = false;
            if (l < 2 || l == 17 || l == 63) printf("lane %2d reg %2d: block(A) %d block(B) %d i %2d j %2d\n", l, r, blkA, blkB, i, j);
        }
    printf("layout D[lane][reg] = block reg/4, row 4*(lane/16) + reg%%4, column lane%%16 : %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    return ok ? 0 : 1;
}
