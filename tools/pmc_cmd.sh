#!/bin/bash
# SQ / cache counters of selected kernels of ANY command:  bash tools/pmc_cmd.sh <tag> <kernel-regex> <rows> -- <command...>
TAG=$1; PAT=$2; ROWS=$3; shift 4
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd $REPO && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err )
done
cd $REPO
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        if not re.search(r"$PAT", k): continue
        key = (k[-44:], row.get("Grid_Size", ""))
        a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for key, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0])[:int("$ROWS")]:
    m = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    print(key, "launches", max(v[1] for v in cs.values()))
    w = max(m.get("SQ_WAVES", 1), 1)
    print("   per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f SMEM %.1f | wave quad-cycles %.0f: wait_any %.0f%% wait_inst %.0f%% active %.0f%%"
          % (m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_VMEM_RD", 0) / w,
             m.get("SQ_INSTS_VMEM_WR", 0) / w, m.get("SQ_INSTS_SMEM", 0) / w, m.get("SQ_WAVE_CYCLES", 0) / w,
             100 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
             100 * m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if gui:
        print("   duration %.0f cycles; VALU quad-cycles per SIMD %.0f (x4 = %.0f%% of duration); waves resident per SIMD %.1f; L2 hit %.2f; LDS conflict %.2f"
              % (gui, m.get("SQ_ACTIVE_INST_VALU", 0) / 1024, 400 * m.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / gui,
                 4 * m.get("SQ_WAVE_CYCLES", 0) / 1024 / gui, m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1),
                 m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
    print("   ", {c: round(v) for c, v in sorted(m.items())})
PY
find $OUT -name "*.csv" -size +1M -delete
