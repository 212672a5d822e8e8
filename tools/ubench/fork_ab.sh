#!/bin/bash
# fork_ab.sh: forked replay graph (searches / network on their own streams) against the single chain, same visit:
# throughput of the 96-fragment job and the single-fragment latency.
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra"
for f in ${FORKS:-0 1 0 1}; do
  D3F_FORK=$f timeout 400 python bench.py --steps ${STEPS:-96} --warmup 8 $LEAN 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork $f', r['value'], r['timing']['p10'], r['timing']['p90'], 'latency', r.get('latency_ms'), 'parity', r.get('parity'))"
done
