#!/bin/bash
# round 6: step A/B of the finalize tails (VT_FIN_TAIL=0: separate launches)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6fintail
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "[VT_FIN_TAIL=$v] " >> "$OUT/step3.log"
    VT_FIN_TAIL=$v timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step3.log" || echo failed >> "$OUT/step3.log"
    echo -n "[vovnet39 VT_FIN_TAIL=$v] " >> "$OUT/step3.log"
    VT_FIN_TAIL=$v timeout -k 10 300 python3 bench.py --model vovnet39 --steps 20 --warmup 6 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step3.log" || echo failed >> "$OUT/step3.log"
  done
done
cat "$OUT/step3.log"
