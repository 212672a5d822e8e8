#!/bin/bash
# x3_knobs.sh <tag> "ENV.." ...: the K-long shapes of tools/gemm_bench.py under each split-plan environment
out=gpurun_out/$1; shift; mkdir -p $out
ONLY="3632,1024,256;905,512,256;905,3840,256;905,768,1024;905,1024,256;198,3840,256;198,256,1024;198,1024,512;198,7680,512;198,1536,2048;905,3072,512;3632,384,512;3632,512,128;3632,256,128"
i=0; cols=""
for v in "$@"; do
  i=$((i+1))
  env $v D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=${SCALE:-5} D3F_GEMM_BENCH_ONLY="$ONLY" timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $out/k_$i.txt
  if [ $i = 1 ]; then cols="<(cut -c1-24 $out/k_$i.txt)"; fi
  cols="$cols <(cut -c32-42 $out/k_$i.txt)"
  echo "$i: $v"
done
eval paste $cols
