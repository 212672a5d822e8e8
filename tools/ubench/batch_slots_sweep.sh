#!/bin/bash
# batch_slots_sweep.sh "B S" ...: the driver's 20-fragment job (bench.py --steps 20 --warmup 5) under each batch x slots split, lean lines
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra --no-latency"
for cfg in "$@"; do set -- $cfg
  timeout 300 python bench.py --steps ${STEPS:-20} --warmup 5 --batch $1 --slots $2 $LEAN 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps ${STEPS:-20} batch $1 slots $2', r['value'], r['timing']['p10'], r['timing']['p90'])"
done
