#!/bin/bash
# First GPU visit of the next round: the prototype and probes that were written after this round's GPU budget was spent.
#   gpurun --timeout 600 -- 'bash tools/next_round.sh r02_v0'
TAG=${1:-next}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for f in gemm_bf16x3 mfma_rate; do
  [ -x tools/ubench/$f.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/$f.hip -o tools/ubench/$f.bin
done
timeout 200 tools/ubench/gemm_bf16x3.bin > $OUT/gemm_bf16x3.txt 2>&1; cat $OUT/gemm_bf16x3.txt
timeout 60 tools/ubench/mfma_rate.bin > $OUT/mfma_rate.txt 2>&1
D3F_GEMM_BENCH_SCALE=4 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_bench.txt; tail -3 $OUT/gemm_bench.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
