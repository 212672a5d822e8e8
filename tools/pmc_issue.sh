#!/bin/bash
# Vector-issue share of every kernel of a replay:  bash tools/pmc_issue.sh <tag> [rows]
# One rocprofv3 --pmc pass over bench.py (F = 4, one slot, timed region only): per kernel and launch size
#   issue = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   -- the fraction of all SIMD cycles spent issuing vector
# instructions (a wave64 vector instruction occupies its SIMD for 4 cycles; packed-f32 / transcendental ones longer, so this is a floor).
# A kernel near 1.0 is bound by its instruction count, whatever its memory pattern looks like (round 6: head32_kernel, 0.93).
TAG=${1:-issue}; ROWS=${2:-40}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra --no-latency"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -- python $REPO/bench.py --steps 8 --warmup 1 --batch 4 --slots 1 $LEAN > /dev/null 2> $OUT/p.err
cd $REPO
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(fn)))
    marks = [int(r["Dispatch_Id"]) for r in rows if "d3f_trace_marker_kernel" in r["Kernel_Name"]]
    lo, hi = (min(marks), max(marks)) if len(set(marks)) >= 2 else (None, None)
    for r in rows:
        if lo is not None and not (lo < int(r["Dispatch_Id"]) < hi):
            continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[-46:], int(r["Grid_Size"]))
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = []
for k, cs in acc.items():
    m = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0: continue
    n = max(v[1] for v in cs.values())
    out.append((cyc * n, k, n, cyc, m))
out.sort(reverse=True)
print("%-46s %9s %5s %9s %7s %9s %9s %8s" % ("kernel", "grid", "n", "us@2.4GHz", "issue", "valu/wave", "salu/wave", "vmem/wave"))
for tot, k, n, cyc, m in out[:int("$ROWS")]:
    w = max(m.get("SQ_WAVES", 1.0), 1.0)
    print("%-46s %9d %5d %9.1f %7.3f %9.0f %9.0f %8.1f" % (k[0], k[1], n, cyc / 2400.0, 4.0 * m.get("SQ_INSTS_VALU", 0.0) / (1024.0 * cyc),
          m.get("SQ_INSTS_VALU", 0.0) / w, m.get("SQ_INSTS_SALU", 0.0) / w, m.get("SQ_INSTS_VMEM_RD", 0.0) / w))
PY
find $OUT -name "*.csv" -size +2M -delete
