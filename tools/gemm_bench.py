#!/usr/bin/env python3
"""Micro-benchmark of d3f_gemm_f32 on the GEMM shapes of one 30k-point self-pair, HIP-event timed.
    python tools/gemm_bench.py            # the library's plan per shape (D3F_GEMM_BENCH_BF16=1: the bf16 contraction)
A/B of kernel variants: build the variant into another .so and point D3FEAT_AMD_LIB at it (tools/gpu_experiments.sh).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_amd import ops  # noqa: E402

SCALE = int(os.environ.get("D3F_GEMM_BENCH_SCALE", "1"))   # fragments per stack (FragmentEngine batch)
N0, N1, N2, N3, N4 = (SCALE * n for n in (58966, 14630, 3632, 905, 198))
# the 26 contraction launches of one replay as the engine issues them since round 3 (fused KPConv forms at levels 0-2, stacked
# resnet branches, decoder contractions on [gathered | skip]): bench.py's roofline.contraction_launches, per fragment
SHAPES = [  # (M, K, N, count)
    (N0, 64, 32, 1), (N0, 96, 128, 1), (N0, 128, 32, 1), (N1, 32, 128, 1), (N1, 128, 64, 1), (N1, 192, 256, 1), (N1, 256, 64, 1),
    (N2, 64, 256, 1), (N2, 256, 128, 1), (N2, 384, 512, 1), (N2, 512, 128, 1), (N3, 128, 512, 1), (N3, 512, 256, 1),
    (N3, 3840, 256, 1), (N3, 768, 1024, 1), (N3, 1024, 256, 1), (N4, 3840, 256, 1), (N4, 256, 1024, 1), (N4, 1024, 512, 1),
    (N4, 7680, 512, 1), (N4, 1536, 2048, 1), (N3, 3072, 512, 1), (N2, 1024, 256, 1), (N1, 512, 128, 1), (N0, 256, 64, 1),
    (N0, 64, 32, 1)]


ONLY = [tuple(int(v) for v in t.split(",")) for t in os.environ.get("D3F_GEMM_BENCH_ONLY", "").split(";") if t]   # "M,K,N;..."
REPS = int(os.environ.get("D3F_GEMM_BENCH_REPS", "0"))


EXTRA = [tuple(int(v) for v in t.split(",")) + (1,) for t in os.environ.get("D3F_GEMM_BENCH_EXTRA", "").split(";") if t]
if EXTRA:
    SHAPES = EXTRA


BF16 = os.environ.get("D3F_GEMM_BENCH_BF16", "0") == "1"     # the bf16-operand contraction (configs[4]) instead of fp32


def time_one(A, B, reps=20):
    if BF16:
        with ops.bf16_contraction():
            return _time_one(A, B, reps)
    return _time_one(A, B, reps)


def _time_one(A, B, reps=20):
    reps = REPS or reps
    for _ in range(1 if REPS else 3):
        ops.gemm(A, B, leaky=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(A, B, leaky=True)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def main():
    dev = torch.device("cuda", 0)
    tot_us, tot_fl = 0.0, 0.0
    maxk = int(os.environ.get("D3F_GEMM_BENCH_MAXK", "0"))
    for (M, K, N, cnt) in SHAPES:
        if maxk and K > maxk:
            continue
        if ONLY and (M // SCALE, K, N) not in ONLY and (M, K, N) not in ONLY:
            continue
        A = torch.randn(M, K, device=dev)
        B = torch.randn(K, N, device=dev)
        fl = 2.0 * M * K * N
        us = time_one(A, B)
        import ctypes
        from d3feat_amd import _lib
        r, c, sl = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        plan = ""
        if K % 32 == 0 and N > 32 and _lib.load().d3f_gemm_x3_plan(M, N, K, 0, ctypes.byref(r), ctypes.byref(c), ctypes.byref(sl)) == 0:
            plan = "x3 %dx%d S=%d" % (r.value, c.value, sl.value)
        gb = 4.0 * M * (K + N) / us / 1e3
        line = "M=%6d K=%5d N=%5d  %-14s %7.1f us %6.1f TF %6.0f GB/s(A+C)" % (M, K, N, plan, us, fl / us / 1e6, gb)
        print(line, flush=True)
        tot_us += us
        tot_fl += fl
    print("total %.1f us for %.2f GFLOP = %.1f TF/s" % (tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))


if __name__ == "__main__":
    main()
