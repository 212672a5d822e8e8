#!/bin/bash
# x3_e2e_knobs.sh "ENV.." ...: lean end-to-end lines (default run) under each environment, alternating with the first
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra --no-latency"
for v in "$@"; do
  env $v timeout 400 python bench.py $LEAN ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['timing']['p10'], r['timing']['p90'])"
done
