#!/bin/bash
# rocprofv3 kernel trace of the bench with N slots: per-kernel totals per fragment (tools/rocpd_summary.py)
TAG=${1:-p1}; SLOTS=${2:-1}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python $REPO/bench.py --steps 64 --warmup 4 --slots $SLOTS --no-cpu-baseline --no-instrument --no-mirror-extra > $OUT/bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*_results.db | head -1)
python $REPO/tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
head -45 $OUT/kernel_stats.csv
find $OUT -name "*.db" -size +20M -delete
