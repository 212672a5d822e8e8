#!/bin/bash
# x3_variants.sh <tag> <variant names...>: tools/gemm_bench.py (F = 5 shapes) for the shipped library and each variant library; X3ENV = extra environment
out=gpurun_out/$1; shift; mkdir -p $out
env $X3ENV D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_x3.txt 2>&1
cols="<(cut -c1-24 $out/g_x3.txt) <(cut -c32-42 $out/g_x3.txt)"
for v in "$@"; do
  env $X3ENV D3FEAT_AMD_LIB=$PWD/d3feat_amd/lib/variants/$v.so D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 120 python tools/gemm_bench.py > $out/g_$v.txt 2>&1
  cols="$cols <(cut -c32-42 $out/g_$v.txt)"
done
echo "shape / x3 $@"
eval paste $cols | grep -v amdgpu.ids
