#!/bin/bash
# SQ counters of the GEMM micro-benchmark, per (kernel, grid):  bash tools/pmc_gemm.sh <tag>   (env of tools/gemm_bench.py applies)
TAG=${1:-pmcg}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export D3F_GEMM_BENCH_REPS=${D3F_GEMM_BENCH_REPS:-2}
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python $REPO/tools/gemm_bench.py > /dev/null 2> $OUT/p1.err
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- python $REPO/tools/gemm_bench.py > /dev/null 2> $OUT/p2.err
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p3 -- python $REPO/tools/gemm_bench.py > /dev/null 2> $OUT/p3.err
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        if "gemm" not in k: continue
        key = (k[-32:], row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
        a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for key, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0])[:24]:
    print(key)
    print("   ", {c.replace("SQ_", ""): round(v[0] / max(v[1], 1)) for c, v in sorted(cs.items())})
PY
find $OUT -name "*.csv" -size +1M -delete
