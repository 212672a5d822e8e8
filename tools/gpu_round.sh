#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats and the two HBM counter passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01_v3 [tests|notests] [pmc|nopmc]'
# Everything lands under gpurun_out/<tag>/ (merged back into the repo's gpurun_out/ by gpurun).
TAG=${1:-run}
TESTS=${2:-tests}
PMC=${3:-pmc}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/box.txt
nproc >> $OUT/box.txt

if [ "$TESTS" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
  echo "pytest exit $?" >> $OUT/tests.log
  tail -5 $OUT/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  tail -2 $OUT/smoke.log
fi

timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err

cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python $REPO/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python $REPO/tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv; head -25 $OUT/kernel_stats.csv; fi

if [ "$PMC" = "pmc" ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra > /dev/null 2> $OUT/pmc_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra > /dev/null 2> $OUT/pmc_write.err
  python $REPO/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/hbm_traffic.json 2> $OUT/pmc_summary.err
  head -c 1500 $OUT/hbm_traffic.json
  # keep the merged output small: the raw per-dispatch CSVs are large
  find $OUT/pmc_fetch $OUT/pmc_write -name "*.csv" -size +2M -delete
fi
rm -rf $OUT/prof/*/*.db-journal
du -sh $OUT
find $OUT -name "*.db" -size +20M -delete
