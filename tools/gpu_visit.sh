#!/bin/bash
# One GPU-box visit, sections chosen by name:  bash tools/gpu_visit.sh <tag> [tests] [smoke] [bench] [prof] [pmc] [proto] [x:<cmd>]
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit.sh r02_v1 tests smoke bench prof'
# Everything lands under gpurun_out/<tag>/ (merged back into the repo's gpurun_out/ by gpurun).
TAG=${1:-run}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; } > $OUT/box.txt
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra"
for SEC in "$@"; do
case "$SEC" in
tests)
  timeout ${TESTS_TIMEOUT:-600} python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -6 $OUT/tests.log;;
tests-all)   # no -x: every failure of a visit in one go
  timeout ${TESTS_TIMEOUT:-600} python -m pytest tests -m gpu -q --durations=30 > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -60 $OUT/tests.log;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
benchk20)   # the driver's command
  timeout ${BENCH_TIMEOUT:-400} python bench.py --steps 20 --warmup 5 --detail-out $OUT/bench_k20.json > $OUT/bench_k20_line.json 2> $OUT/bench_k20.err; echo "bench exit $?" >> $OUT/bench_k20.err
  wc -c $OUT/bench_k20_line.json; cut -c1-400 $OUT/bench_k20_line.json;;
bench)
  # stdout = the compact driver-facing line (bench_line.json); the full object goes to bench.json (as in earlier rounds)
  timeout ${BENCH_TIMEOUT:-400} python bench.py ${BENCH_FLAGS:-} --detail-out $OUT/bench.json > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
  python - <<PY
import json
try:
    raw = open("$OUT/bench_line.json").read()
    l = json.loads(raw.strip().splitlines()[-1])
    print("stdout bytes", len(raw), {k: l[k] for k in ("value", "ms_per_step", "parity", "vs_cpu_baseline", "latency_ms")})
    print(l["roofline"])
    r = json.load(open("$OUT/bench.json"))
    print(r["roofline"]["timed_kernels_ms_per_step"])
    print(r["cpu_baseline"])
except Exception as e:
    print("bench line unreadable:", e)
PY
  tail -3 $OUT/bench.err;;
benchlean)
  timeout 600 python bench.py $LEAN > $OUT/bench_lean.json 2> $OUT/bench_lean.err; cut -c1-160 $OUT/bench_lean.json;;
prof)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python $REPO/bench.py --steps ${PROF_STEPS:-64} --warmup 8 ${PROF_FLAGS:-} $LEAN > $OUT/prof_bench.json 2> $OUT/prof.err
  cd $REPO
  DB=$(ls $OUT/prof/*/*_results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    python tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
    python tools/rocpd_summary.py --timed-region --fragments ${PROF_STEPS:-64} $DB > $OUT/kernel_stats_timed_region.csv
    head -32 $OUT/kernel_stats_timed_region.csv | cut -c1-150
    python tools/timeline_summary.py $DB > $OUT/timeline.json 2>/dev/null
  fi
  rm -rf $OUT/prof/*/*.db-journal; find $OUT -name "*.db" -size +20M -delete;;
prof1)   # ONE replay in flight: kernel durations without the other streams' kernels competing for the CUs (with several
         # replays in flight a kernel's duration in the trace includes the time its workgroups wait for a free CU)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof1 -- python $REPO/bench.py --steps 32 --warmup 8 --slots 1 --batch 4 $LEAN > $OUT/prof1_bench.json 2> $OUT/prof1.err
  cd $REPO
  DB=$(ls $OUT/prof1/*/*_results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    python tools/rocpd_summary.py --timed-region --fragments 32 $DB > $OUT/kernel_stats_one_replay_in_flight.csv
    head -40 $OUT/kernel_stats_one_replay_in_flight.csv | cut -c1-150
    python tools/gs_timeline.py $DB 2 34
    [ -n "$PROF1_BYGRID" ] && python tools/kernel_by_grid.py $DB "$PROF1_BYGRID"
  fi
  rm -rf $OUT/prof1/*/*.db-journal; find $OUT -name "*.db" -size +20M -delete;;
pmc)
  cd /tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 16 --warmup 1 --batch 4 $LEAN --no-latency > /dev/null 2> $OUT/pmc_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --steps 16 --warmup 1 --batch 4 $LEAN --no-latency > /dev/null 2> $OUT/pmc_write.err
  cd $REPO
  python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/hbm_traffic.json 2> $OUT/pmc_summary.err
  head -c 600 $OUT/hbm_traffic.json
  find $OUT/pmc_fetch $OUT/pmc_write -name "*.csv" -size +2M -delete;;
mfma)   # matrix-pipe counters of the MFMA kernels (ONE --pmc pass with --kernel-trace only; tools/pmc_mfma.py), at the headline's
        # F = 12 fragments per replay by default so that the family figures are comparable with the line's issued_tflops / 2516.6
        # (MFMA_BATCH=4: the F = 4 shapes of the one-replay tables)
  cd /tmp
  MB=${MFMA_BATCH:-12}
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python $REPO/bench.py --steps $((MB * 2)) --warmup 1 --batch $MB --slots 1 $LEAN --no-latency > /dev/null 2> $OUT/pmc_mfma.err
  cd $REPO
  D3F_PMC_FRAGMENTS_PER_LAUNCH=$MB python tools/pmc_mfma.py $OUT/pmc_mfma > $OUT/mfma_counters.json 2> $OUT/pmc_mfma_summary.err
  head -c 1500 $OUT/mfma_counters.json; tail -3 $OUT/pmc_mfma.err
  find $OUT/pmc_mfma -name "*.csv" -size +2M -delete;;
sq:*)   # SQ / cache counters of selected kernels on the graph engine's F = 4 shapes:  sq:<kernel-regex>
  BENCH_ARGS="--steps 8 --warmup 1 --batch 4 --slots 1 $LEAN" bash tools/pmc_kernel.sh $TAG/sq "${SEC#sq:}" 12 > $OUT/sq_counters.txt 2>&1; cat $OUT/sq_counters.txt | cut -c1-600;;
proto)
  for f in gemm_bf16x3 mfma_rate; do
    [ -x tools/ubench/$f.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/$f.hip -o tools/ubench/$f.bin
  done
  timeout 200 tools/ubench/gemm_bf16x3.bin > $OUT/gemm_bf16x3.txt 2>&1; cat $OUT/gemm_bf16x3.txt;;
x:*)
  CMD="${SEC#x:}"; echo "== $CMD"; timeout 900 bash -c "$CMD" 2>&1 | tail -40;;
esac
done
du -sh $OUT
