#!/bin/bash
# A/B measurements in ONE GPU visit: every line = one lean bench.py run (fragments/s) under an environment / flag variation.
#   gpurun --timeout 900 -- 'bash tools/gpu_experiments.sh r02_x1 "<name>|<env assignments>|<bench flags>" ...'
TAG=${1:-exp}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
LEAN="--no-cpu-baseline --no-instrument --no-mirror-extra --no-pcie-extra"
: > $OUT/experiments.txt
for SPEC in "$@"; do
  NAME="${SPEC%%|*}"; REST="${SPEC#*|}"; ENVS="${REST%%|*}"; FLAGS="${REST#*|}"
  LINE=$(env $ENVS timeout ${RUN_TIMEOUT:-90} python bench.py $LEAN $FLAGS 2> $OUT/$NAME.err | tail -1)
  VAL=$(python -c "import json,sys; r=json.loads(sys.argv[1]); print('%.1f fragments/s  %.4f ms/step  fallbacks=%s' % (r['value'], r['ms_per_step'], r.get('config',{}).get('engine_fallbacks')))" "$LINE" 2>/dev/null || echo "FAILED: $(tail -2 $OUT/$NAME.err | tr '\n' ' ')")
  printf "%-28s %-40s %-28s %s\n" "$NAME" "$ENVS" "$FLAGS" "$VAL" | tee -a $OUT/experiments.txt
done
