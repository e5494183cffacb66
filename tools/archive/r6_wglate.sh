#!/bin/bash
# round 6: every filter gradient released behind its unit's data gradient (VT_WGRAD_LATE=1) instead of in front of it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6wglate
mkdir -p "$OUT"; cd "$ROOT"; rm -f "$OUT/step.log"
for rep in 1 2 3; do
  for v in 0 1; do
    for model in cspdarknet53 vovnet39; do
      echo -n "[$model VT_WGRAD_LATE=$v] " >> "$OUT/step.log"
      VT_WGRAD_LATE=$v timeout -k 10 300 python3 bench.py --model $model --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
    done
  done
done
cat "$OUT/step.log"
