#!/bin/bash
# SQ counters of the eager bench for selected kernels:  bash tools/pmc_kernel.sh <tag> <kernel-regex> [rows]
#   BENCH_ARGS overrides the bench flags (default: eager F = 1 shapes), e.g. the graph engine's F = 4 shapes:
#   BENCH_ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra"
TAG=${1:-pmc}; PAT=${2:-nb_search}
BA=${BENCH_ARGS:---steps 3 --warmup 1 --eager --no-cpu-baseline --no-instrument}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python $REPO/bench.py $BA > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python $REPO/bench.py $BA > /dev/null 2> $OUT/p2.err
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -- python $REPO/bench.py $BA > /dev/null 2> $OUT/p3.err
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        if not re.search(r"$PAT", k): continue
        key = (k[-40:], row.get("Grid_Size", ""))
        a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for key, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0])[:int("${3:-8}")]:
    print(key)
    print("   ", {c: round(v[0] / max(v[1], 1)) for c, v in sorted(cs.items())})
PY
find $OUT -name "*.csv" -size +1M -delete
