#!/bin/bash
# x3_variants.sh <tag> <variant names...>: tools/gemm_bench.py for the shipped library and each variant library (X3SHAPES: "M,K,N;..." instead of the network's list)
out=gpurun_out/$1; shift; mkdir -p $out
export D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=${SCALE:-5}
[ -n "$X3SHAPES" ] && export D3F_GEMM_BENCH_EXTRA="$X3SHAPES" D3F_GEMM_BENCH_SCALE=1
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $out/g_x3.txt
cols="<(cut -c1-24 $out/g_x3.txt) <(cut -c32-42 $out/g_x3.txt)"
for v in "$@"; do
  D3FEAT_AMD_LIB=$PWD/d3feat_amd/lib/variants/$v.so timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $out/g_$v.txt
  cols="$cols <(cut -c32-42 $out/g_$v.txt)"
done
echo "shape / x3 $@"
eval paste $cols
