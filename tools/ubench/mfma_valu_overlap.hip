// Micro-benchmark: does vector-ALU work hide behind v_mfma_f32_32x32x16_bf16 on gfx950 -- inside one wavefront (interleaved by the
// compiler / by sched_group_barrier) and across the wavefronts of a SIMD?  Per loop iteration and wave: NM MFMAs on two accumulator
// chains + NV vector instructions of the operand-split flavour (v_cvt_pk_bf16_f32, shift, and, subtract), independent of the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sub1(float a, float b) {       // one v_sub_f32 the SLP vectoriser cannot pack into v_pk_add_f32
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// MODE 0: MFMAs only.  1: VALU only.  2: both, VALU group after the MFMA group (program order).  3: both, interleaved 1 MFMA : NV/NM VALU
template <int NM, int NVG, int MODE>     // NVG: groups of 5 vector instructions (cvt, shl, and, sub, sub)
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    uint4 a = make_uint4(threadIdx.x, 2, 3, 4), b = make_uint4(5, 6, threadIdx.x, 8);
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + i + threadIdx.x; y[i] = seed * 3 + i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
#pragma unroll
                for (int g = m * NVG / NM; g < (m + 1) * NVG / NM; ++g) {
                    const int i = g & 7;
                    const unsigned p = cvt_pk(x[i], y[i]);
                    x[i] = sub1(x[i], __uint_as_float(p << 16));
                    y[i] = sub1(y[i], __uint_as_float(p & 0xffff0000u));
                }
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 5 * (NVG / NM), 0);
            }
        } else {
            if (MODE != 1) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                }
            }
            if (MODE != 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NVG; ++g) {
                    const int i = g & 7;
                    const unsigned p = cvt_pk(x[i], y[i]);
                    x[i] -= __uint_as_float(p << 16);
                    y[i] -= __uint_as_float(p & 0xffff0000u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NVG, int MODE>
static void run(const char* name, int blocks, float* out) {
    const int iters = 2048;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<NM, NVG, MODE><<<blocks, 256>>>(out, 16, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<NM, NVG, MODE><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const int wps = (blocks + 255) / 256;      // waves per SIMD
    printf("%-52s %d wave(s)/SIMD: %8.3f ms = %7.0f ns per iteration and SIMD (%d MFMA = %d cycles of matrix pipe, %d VALU)\n", name, wps, ms,
           ms * 1e6 / iters, NM * wps, NM * wps * 32, (MODE == 0 ? 0 : NVG * 5) * wps);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    for (int blocks = 256; blocks <= 768; blocks += 256) {
        run<24, 16, 0>("24 MFMA", blocks, out);
        run<24, 24, 1>("120 VALU", blocks, out);
        run<24, 24, 2>("24 MFMA then 120 VALU (program order)", blocks, out);
        run<24, 24, 3>("24 MFMA interleaved with 120 VALU (1 : 5), scalar subs", blocks, out);
        run<24, 16, 3>("24 MFMA interleaved with 80 VALU (3 : 10), scalar subs", blocks, out);
        run<24, 48, 3>("24 MFMA interleaved with 240 VALU (1 : 10), scalar subs", blocks, out);
    }
    return 0;
}
