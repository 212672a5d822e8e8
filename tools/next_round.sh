#!/bin/bash
# First GPU visit of the next round: the measurements this round's GPU budget no longer covered, on the final code.
#   gpurun --timeout 900 -- 'bash tools/next_round.sh r03_v0'
TAG=${1:-next}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
# 1. the configs[4] line (bf16 contraction) with the sort form of the stage-0 subsampling in place
timeout 300 python bench.py --bf16 --batch 8 --slots 3 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; cut -c1-200 $OUT/bench_bf16.json
# 2. the driver's command, all legs
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cut -c1-200 $OUT/bench_k20.json
# 3. what the sort form costs / saves: geometry-only regime, sort against hash (D3F_GS_SORT_MIN=0), and the full regime
G="--ablate gemm,kpconv,rowpos,maxpool,head,pack"
RUN_TIMEOUT=60 bash tools/gpu_experiments.sh ${TAG}_x "sort||" "hash|D3F_GS_SORT_MIN=0|" "sort_geo||$G" "hash_geo|D3F_GS_SORT_MIN=0|$G" \
    "sort_k20||--steps 20 --warmup 5" "hash_k20|D3F_GS_SORT_MIN=0|--steps 20 --warmup 5"
# 4. SQ counters of the rocPRIM digit passes at the engine's shapes
BENCH_ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra" \
    bash tools/pmc_kernel.sh ${TAG}_pmc "onesweep|gs_sortkey|gs_voxels" 6 > $OUT/pmc_sort.txt 2>&1; tail -12 $OUT/pmc_sort.txt | cut -c1-400
