#!/bin/bash
# x3_visit.sh <tag> ["ENV=.. ENV=.." ...]: unit tests of the operand-split contraction, then tools/gemm_bench.py (F = 5 shapes): the fp32
# MFMA kernel, the shipped x3 kernel, and the x3 kernel under each extra environment (plan knobs)
out=gpurun_out/$1; shift; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_gemm_x3.py -x -q > $out/x3_tests.log 2>&1; echo "x3 tests rc $?"; tail -3 $out/x3_tests.log
timeout 300 python -m pytest tests/test_gpu_network.py -x -q -k gemm > $out/net_gemm_tests.log 2>&1; echo "network gemm tests rc $?"; tail -2 $out/net_gemm_tests.log
D3F_GEMM_X3=0 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_fp32.txt 2>&1
D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_x3.txt 2>&1
cols="<(cut -c1-24 $out/g_fp32.txt) <(cut -c32-42 $out/g_fp32.txt) <(cut -c32-42 $out/g_x3.txt)"
i=0
for v in "$@"; do
  i=$((i+1))
  env $v D3F_GEMM_X3=1 D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/g_env$i.txt 2>&1
  cols="$cols <(cut -c32-42 $out/g_env$i.txt)"
done
echo "shape / fp32 x3 $@"
eval paste $cols | grep -v amdgpu.ids
