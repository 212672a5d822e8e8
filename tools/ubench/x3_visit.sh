#!/bin/bash
# one GPU visit for the operand-split contraction: unit tests, then the launch-by-launch A/B (tools/gemm_bench.py, F = 5 shapes)
out=gpurun_out/$1; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s > $out/x3_tests.log 2>&1; echo "x3 tests rc $?" | tee -a $out/x3_tests.log
tail -25 $out/x3_tests.log
timeout 300 python -m pytest tests/test_gpu_network.py -x -q -k gemm > $out/net_gemm_tests.log 2>&1; echo "network gemm tests rc $?"; tail -3 $out/net_gemm_tests.log
for x in 0 1; do
  D3F_GEMM_X3=$x D3F_GEMM_BENCH_SCALE=5 timeout 300 python tools/gemm_bench.py > $out/gemm_x3_$x.txt 2>&1
  tail -1 $out/gemm_x3_$x.txt
done
paste <(cut -c1-60 $out/gemm_x3_0.txt) <(cut -c28-60 $out/gemm_x3_1.txt) | tail -28
