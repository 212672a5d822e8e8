#!/bin/bash
# x3_knobs.sh <tag> "ENV.." ...: tools/gemm_bench.py (F = SCALE shapes, all 26 launches) under each environment
out=gpurun_out/$1; shift; mkdir -p $out
i=0; cols=""
for v in "$@"; do
  i=$((i+1))
  env D3F_GEMM_X3=1 $v D3F_GEMM_BENCH_SCALE=${SCALE:-5} timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $out/k_$i.txt
  if [ $i = 1 ]; then cols="<(cut -c1-24 $out/k_$i.txt)"; fi
  cols="$cols <(cut -c32-42 $out/k_$i.txt)"
  echo "$i: $v"
done
eval paste $cols
