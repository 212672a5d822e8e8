// Prototype for the next round: fp32-accurate GEMM on the bf16 matrix cores by operand splitting.
//   a = a1 + a2 + a3 (three bf16 pieces, 8 significant bits each, residuals exact in fp32), same for b;
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1): six v_mfma_f32_32x32x16_bf16 per 32x32x16 block (32 cycles each)
//   instead of eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.67x the matrix-pipe rate at ~2^-22 relative product error.
// Standalone: builds its own operands, checks sampled rows against a float64 host reference, times the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_bf16x3.hip -o tools/ubench/gemm_bf16x3.bin && tools/ubench/gemm_bf16x3.bin
// NOT part of libd3feat_amd.so and not validated on hardware yet (written after the round's GPU budget was spent).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BM 64
#define BN 64
#define BK 32
#define LROW 40                       // bf16 per LDS row: 32 + 8 pad = 80 bytes (16-byte aligned, bank-spread for b128 reads)
#define PLANE (64 * LROW)             // one split plane of a 64-row tile, in bf16 units

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// eight consecutive-k floats -> three planes of eight bf16 (one uint4 each)
template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], uint4 (&pl)[3]) {
    unsigned p[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        p[0][i] = cvt_pk_bf16(a, b);
        a -= bf_lo(p[0][i]); b -= bf_hi(p[0][i]);
        p[1][i] = NS > 1 ? cvt_pk_bf16(a, b) : 0u;
        a -= bf_lo(p[1][i]); b -= bf_hi(p[1][i]);
        p[2][i] = NS > 2 ? cvt_pk_bf16(a, b) : 0u;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) pl[s] = make_uint4(p[s][0], p[s][1], p[s][2], p[s][3]);
}

// C[M,N] = A[M,K] @ B[K,N], all fp32 row-major; K % 32 == 0, N % 64 == 0 (prototype), any M.
template <int NS>   // pieces per operand: 1 (plain bf16), 2 (3 products), 3 (6 products)
__global__ void __launch_bounds__(256) gemm_bf16split_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                             int ldb, float* __restrict__ C, int ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];   // [2 buffers][A planes 3 | B planes 3][64][LROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // staging slots: A: row tid/4, k segment 8*(tid%4) (two float4);  B: column tid%64, k segment 8*(tid/64) (eight dwords,
    // each coalesced across the 64 lanes of a wavefront)
    const int ar = tid >> 2, ak = (tid & 3) << 3;
    const int bn = tid & 63, bk = (tid >> 6) << 3;
    const int arow = min(m0 + ar, M - 1);          // clamped: rows past M are never stored
    const float* ap = A + (size_t)arow * lda + ak;
    const float* bp = B + (size_t)bk * ldb + n0 + bn;
    float4 va0, va1;      // kept as loaded: unpacking them here would make the compiler wait for the loads at once
    float xb[8];
    auto load_tile = [&](int t) {
        va0 = *(const float4*)(ap + t * BK);
        va1 = *(const float4*)(ap + t * BK + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[e] = bp[(size_t)(t * BK + e) * ldb];
    };
    auto store_tile = [&](int buf) {
        __bf16* base = smem + buf * 6 * PLANE;
        uint4 pl[3];
        const float xa[8] = {va0.x, va0.y, va0.z, va0.w, va1.x, va1.y, va1.z, va1.w};
        split8<NS>(xa, pl);
#pragma unroll
        for (int s = 0; s < NS; ++s) *(uint4*)&base[s * PLANE + ar * LROW + ak] = pl[s];
        split8<NS>(xb, pl);
#pragma unroll
        for (int s = 0; s < NS; ++s) *(uint4*)&base[(3 + s) * PLANE + bn * LROW + bk] = pl[s];
    };
    f32x16 hi, lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) { hi[i] = 0.f; lo[i] = 0.f; }
    const int nt = K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int afrag = (wm * 32 + (lane & 31)) * LROW + (lane >> 5) * 8;
    const int bfrag = (wn * 32 + (lane & 31)) * LROW + (lane >> 5) * 8;
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);
        const __bf16* base = smem + cur * 6 * PLANE;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fa[kb][s] = *(const bf16x8*)&base[s * PLANE + afrag + 16 * kb];
                fb[kb][s] = *(const bf16x8*)&base[(3 + s) * PLANE + bfrag + 16 * kb];
            }
        __builtin_amdgcn_sched_barrier(0);
        // operands swapped (B fragment first): the accumulator holds the transposed tile -> 16-byte stores
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][0], fa[kb][0], hi, 0, 0, 0);
            if (NS > 1) {
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][1], fa[kb][0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][0], fa[kb][1], lo, 0, 0, 0);
            }
            if (NS > 2) {
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][2], fa[kb][0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][1], fa[kb][1], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb][0], fa[kb][2], lo, 0, 0, 0);
            }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // transposed accumulator: lane = output row (lane & 31) of the wave's tile, register quad q = columns 8q + 4(lane>>5) .. +3
    const int gm = m0 + wm * 32 + (lane & 31);
    float4 o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = hi[4 * q + e] + lo[4 * q + e];
            asm volatile("" : "+v"(v[e]));
        }
        o[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
    float* dst = C + (size_t)(gm < M ? gm : 0) * ldc + n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (gm < M) *(float4*)&dst[8 * q] = o[q];
}

template <int NS>
static void run(const char* name, int M, int K, int N, const float* dA, const float* dB, float* dC, const std::vector<float>& hA,
                const std::vector<float>& hB) {
    const size_t lds = (size_t)2 * 6 * PLANE * sizeof(__bf16);
    hipFuncSetAttribute((const void*)gemm_bf16split_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(N / BN, (M + BM - 1) / BM);
    gemm_bf16split_kernel<NS><<<grid, 256, lds>>>(dA, K, dB, N, dC, N, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int reps = 20;
    hipEventRecord(s);
    for (int i = 0; i < reps; ++i) gemm_bf16split_kernel<NS><<<grid, 256, lds>>>(dA, K, dB, N, dC, N, M, N, K);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    ms /= reps;
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * sizeof(float), hipMemcpyDeviceToHost);
    // float64 reference on 64 sampled rows
    double maxerr = 0.0, maxref = 0.0;
    for (int si = 0; si < 64; ++si) {
        const int m = (int)(((long long)si * 2654435761ll) % M);
        for (int n = 0; n < N; ++n) {
            double r = 0.0;
            for (int k = 0; k < K; ++k) r += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)k * N + n];
            maxerr = fmax(maxerr, fabs(r - (double)hC[(size_t)m * N + n]));
            maxref = fmax(maxref, fabs(r));
        }
    }
    printf("%-26s M=%6d K=%5d N=%5d: %8.1f us  %7.1f TF/s (fp32-equivalent)  max err / max |C| = %.2e\n", name, M, K, N,
           ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr / maxref);
}

int main() {
    const int shapes[][3] = {{234956, 64, 128}, {234956, 256, 64}, {58320, 960, 64}, {58320, 512, 128}, {14528, 1920, 128},
                             {3620, 3840, 256}, {788, 7680, 512}, {4096, 4096, 4096}};
    for (auto& sh : shapes) {
        const int M = sh[0], K = sh[1], N = sh[2];
        std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
        unsigned st = 12345u + M + K + N;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = rnd() * (1.0f + 3.0f * rnd() * rnd());
        for (auto& v : hB) v = rnd() / sqrtf((float)K);
        float *dA, *dB, *dC;
        hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run<1>("bf16 (1 product)", M, K, N, dA, dB, dC, hA, hB);
        run<2>("bf16x2 (3 products)", M, K, N, dA, dB, dC, hA, hB);
        run<3>("bf16x3 (6 products)", M, K, N, dA, dB, dC, hA, hB);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
