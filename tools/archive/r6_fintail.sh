#!/bin/bash
# round 6: the BatchNorm finalize step as the tail of the producing launch -- parity + step A/B (VT_FIN_TAIL=0: separate launches)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6fintail
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_fin_tail_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a "$OUT/tests.log"
grep -q passed "$OUT/tests.log" && ! grep -q failed "$OUT/tests.log" || exit 1
timeout -k 10 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a "$OUT/tests.log"
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "[VT_FIN_TAIL=$v] " >> "$OUT/step.log"
    VT_FIN_TAIL=$v timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
    echo -n "[vovnet39 VT_FIN_TAIL=$v] " >> "$OUT/step.log"
    VT_FIN_TAIL=$v timeout -k 10 300 python3 bench.py --model vovnet39 --steps 20 --warmup 6 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
