#!/bin/bash
# rocprofv3 kernel trace of the bench: per-kernel totals (tools/rocpd_summary.py)
#   bash tools/prof_one.sh <tag> [slots] [batch]      (batch 16, slots 1: kernels run alone -> the work-bound breakdown)
TAG=${1:-p1}; SLOTS=${2:-1}; BATCH=${3:-4}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python $REPO/bench.py --steps $((BATCH * 8)) --warmup 4 --slots $SLOTS --batch $BATCH --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra > $OUT/bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*_results.db | head -1)
python $REPO/tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
head -${4:-45} $OUT/kernel_stats.csv
find $OUT -name "*.db" -size +20M -delete
