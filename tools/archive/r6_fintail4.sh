#!/bin/bash
# round 6: forward finalize tails (span6) -- parity, trainer tests, step A/B (VT_FIN_TAIL=0 | backward tails only | both)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6fintail
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/tests4.log" "$OUT/step4.log"
timeout -k 10 600 python -m pytest tests/test_fin_tail_gpu.py tests/test_span6_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a "$OUT/tests4.log"
grep -q passed "$OUT/tests4.log" && ! grep -q failed "$OUT/tests4.log" || exit 1
run() { # label, env...
  echo -n "[$1] " >> "$OUT/step4.log"; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step4.log" || echo failed >> "$OUT/step4.log"
}
for rep in 1 2 3; do
  run "separate launches" VT_FIN_TAIL=0
  run "backward tails only" VT_FIN_TAIL=1 VT_FIN_TAIL_SPAN6=0
  run "backward + span6 forward tails" VT_FIN_TAIL=1
done
cat "$OUT/step4.log"
timeout -k 10 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a "$OUT/tests4.log"
