#!/bin/bash
# round 6: the pointwise apply passes finalize for themselves -- parity + trainer tests + step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6pwfin
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/tests.log" "$OUT/step.log"
timeout -k 10 600 python -m pytest tests/test_pointwise_gpu.py tests/test_bn_fin_apply_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a "$OUT/tests.log"
grep -q passed "$OUT/tests.log" && ! grep -q failed "$OUT/tests.log" || exit 1
run() { # label, model, env...
  echo -n "[$1] " >> "$OUT/step.log"; model=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --model $model --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2 3; do
  run "separate finalize launches" cspdarknet53 VT_BN_FIN_APPLY=0
  run "finalize inside the passes" cspdarknet53 VT_BN_FIN_APPLY=1
  run "vovnet39: separate finalize launches" vovnet39 VT_BN_FIN_APPLY=0
  run "vovnet39: finalize inside the passes" vovnet39 VT_BN_FIN_APPLY=1
done
cat "$OUT/step.log"
timeout -k 10 900 python -m pytest tests/test_trainer_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a "$OUT/tests.log"
