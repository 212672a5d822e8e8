#!/bin/bash
# x3_variants.sh <tag> <variant names...>: tools/gemm_bench.py (F = 5 shapes) per variant library, one table
out=gpurun_out/$1; shift; mkdir -p $out
D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_base.txt 2>&1
cols="<(cut -c1-47 $out/g_base.txt)"
for v in "$@"; do
  D3FEAT_AMD_LIB=$PWD/d3feat_amd/lib/variants/$v.so D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_$v.txt 2>&1
  cols="$cols <(cut -c32-42 $out/g_$v.txt)"
done
echo "shape / base $@"
eval paste $cols | grep -v amdgpu.ids
