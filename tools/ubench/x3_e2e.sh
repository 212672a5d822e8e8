#!/bin/bash
# x3_e2e.sh <tag>: end-to-end A/B of the operand-split contraction inside one visit (lean bench lines), then kernel durations with one replay in flight
out=gpurun_out/$1; mkdir -p $out
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra --no-latency"
for x in 0 1 0 1; do
  D3F_GEMM_X3=$x timeout 400 python bench.py $LEAN > $out/bench_x3_$x.json 2> $out/bench_x3_$x.err
  python - <<PY
import json
r = json.loads(open("$out/bench_x3_$x.json").read().strip().splitlines()[-1])
print("x3=$x", r["value"], r["timing"]["p10"], r["timing"]["p90"], r["parity"])
PY
done
