#!/bin/bash
# round 6, GPU call K: finalize inside the consuming launch -- parity, model tests, step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r6k}
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python -m pytest tests/test_bn_fin_apply_gpu.py -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?; echo "kernel tests exit $rc" | tee -a "$OUT/status.txt"; tail -15 "$OUT/tests.log" | cut -c1-250
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests/test_trainer_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "frozen_bn or first_step or grouped or block_vs_reference or train_steps_f32 or bf16_train_step or deterministic_mode_gives or validation" > "$OUT/tests2.log" 2>&1
rc=$?; echo "model tests exit $rc" | tee -a "$OUT/status.txt"; tail -3 "$OUT/tests2.log" | cut -c1-250
for rep in 1 2 3; do
  for cfg in "VT_BN_FIN_APPLY=0" "VT_BN_FIN_APPLY=1" "VT_BN_FIN_APPLY=1 VT_SPAN6_MASK=2"; do
    echo -n "[$cfg] " >> "$OUT/step.log"
    env $cfg timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
for cfg in "VT_BN_FIN_APPLY=0" "VT_BN_FIN_APPLY=1"; do
  echo -n "[batch 128 $cfg] " >> "$OUT/step128.log"
  env $cfg timeout -k 10 300 python3 bench.py --batch 128 --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step128.log" || echo failed >> "$OUT/step128.log"
  echo -n "[vovnet39 $cfg] " >> "$OUT/step128.log"
  env $cfg timeout -k 10 300 python3 bench.py --model vovnet39 --steps 20 --warmup 6 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step128.log" || echo failed >> "$OUT/step128.log"
done
cat "$OUT/step128.log"
