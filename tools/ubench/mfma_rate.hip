// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 under the operand patterns of the GEMM kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>   // MODE 0: operands in registers; 1: operands re-read from LDS (ds_read_b128) every 8 k-steps
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 36 * 2];
    for (int i = threadIdx.x; i < 64 * 36 * 2; i += 256) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int u = 0; u < NACC; ++u)
        for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    const int lane = threadIdx.x & 63;
    float4 a0 = make_float4(seed, seed + 1, seed + 2, seed + 3), a1 = a0, b0 = a0, b1 = a0;
    const float* pa = &lds[(lane & 31) * 36 + (lane >> 5) * 16];
    const float* pb = &lds[64 * 36 + (lane & 31) * 36 + (lane >> 5) * 16];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            a0 = *(const float4*)(pa + (it & 1) * 8); a1 = *(const float4*)(pa + (it & 1) * 8 + 4);
            b0 = *(const float4*)(pb + (it & 1) * 8); b1 = *(const float4*)(pb + (it & 1) * 8 + 4);
        }
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[u], 0, 0, 0);
    }
    float s = 0.f;
    for (int u = 0; u < NACC; ++u)
        for (int i = 0; i < 16; ++i) s += acc[u][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
static void run(const char* name, int blocks, float* out) {
    const int iters = 4096;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<NACC, MODE><<<blocks, 256>>>(out, 16, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<NACC, MODE><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double mfma_per_wave = (double)iters * 8 * NACC;
    const double flops = mfma_per_wave * 4096.0 * blocks * 4;
    printf("%-34s blocks %5d: %8.3f ms  %7.1f TF/s  %6.1f ns per MFMA per wave-slot\n", name, blocks, ms, flops / ms / 1e9,
           ms * 1e6 / mfma_per_wave / ((blocks + 255) / 256));
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    run<1, 0>("1 acc, register operands", 256, out);
    run<4, 0>("4 acc, register operands", 256, out);
    run<1, 1>("1 acc, LDS operands", 256, out);
    run<4, 1>("4 acc, LDS operands", 256, out);
    run<1, 0>("1 acc, reg, 4 waves/SIMD", 1024, out);
    run<1, 1>("1 acc, LDS, 4 waves/SIMD", 1024, out);
    run<2, 1>("2 acc, LDS, 2 waves/SIMD", 512, out);
    return 0;
}
