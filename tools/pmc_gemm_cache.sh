TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export D3F_GEMM_BENCH_REPS=2
i=0
for set in "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/q$i -- python $REPO/tools/gemm_bench.py > /dev/null 2> $OUT/q$i.err || echo "set $i failed: $(grep -i -m2 'error\|invalid\|not' $OUT/q$i.err | cut -c1-200)"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/q*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        if "gemm" not in k: continue
        key = (k[-32:], row.get("Grid_Size", ""))
        a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for key, cs in sorted(acc.items()):
    print(key, {c: round(v[0] / max(v[1], 1)) for c, v in sorted(cs.items())})
PY
find $OUT -name "*.csv" -size +1M -delete
