// Go / no-go probe for a sort-based stage-0 subsampler: rocPRIM radix_sort_pairs of N (key, index) pairs with `bits` key bits,
// timed stand-alone and replayed from a captured HIP graph.   hipcc --offload-arch=gfx950 -O3 radix_sort_probe.hip -o radix_sort_probe.bin
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill_kernel(unsigned* k, unsigned* v, int n, unsigned mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned x = (unsigned)i * 2654435761u; x ^= x >> 13; x *= 0x9E3779B1u; k[i] = (x >> 4) & mask; v[i] = i; }
}

int main() {
    const int sizes[] = {1264096};
    const int bitsv[] = {32};
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int n : sizes) for (int bits : bitsv) {
        unsigned *k0, *k1, *v0, *v1; void* tmp = nullptr; size_t tb = 0;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
        CK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, (size_t)n, 0, bits, st));
        CK(hipMalloc(&tmp, tb));
        const unsigned mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        fill_kernel<<<(n + 255) / 256, 256, 0, st>>>(k0, v0, n, mask);
        CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)n, 0, bits, st));
        CK(hipStreamSynchronize(st));
        // correctness: sorted + stable
        std::vector<unsigned> hk(n), hv(n);
        CK(hipMemcpy(hk.data(), k1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv.data(), v1, n * 4, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int i = 1; i < n && ok; ++i) ok = hk[i - 1] < hk[i] || (hk[i - 1] == hk[i] && hv[i - 1] < hv[i]);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int r = 0; r < 20; ++r) CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)n, 0, bits, st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        // graph capture
        hipGraph_t g; hipGraphExec_t ge; float gms = -1.f;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        fill_kernel<<<(n + 255) / 256, 256, 0, st>>>(k0, v0, n, mask);      // inputs rewritten by every replay, as in the library
        hipError_t ce = rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)n, 0, bits, st);
        hipError_t ee = hipStreamEndCapture(st, &g);
        bool gok = false;
        if (ce == hipSuccess && ee == hipSuccess && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(a, st));
            for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&gms, a, b)); gms /= 20.f;
            CK(hipMemcpy(hk.data(), k1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv.data(), v1, n * 4, hipMemcpyDeviceToHost));
            gok = true;
            for (int i = 1; i < n && gok; ++i) gok = hk[i - 1] < hk[i] || (hk[i - 1] == hk[i] && hv[i - 1] < hv[i]);
        } else printf("capture failed: %s / %s\n", hipGetErrorString(ce), hipGetErrorString(ee));
        printf("after 21 graph replays sorted+stable=%d\n", (int)gok);
        printf("n=%8d bits=%2d  sorted+stable=%d  temp=%zu B  %.1f us per sort, %.1f us per graph replay\n", n, bits, (int)ok, tb, ms / 20.f * 1e3f, gms * 1e3f);
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
    }
    return 0;
}
