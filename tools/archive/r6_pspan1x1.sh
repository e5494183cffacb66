#!/bin/bash
# round 6: the persistent resident-filter kernel on the 1x1 layers with <= 128 output channels (VT_PSPAN=2 lifts its "ntaps >= 4"
# and "M >= 262144" conditions) against the input-span kernel they run on
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6pspan1x1
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/ab.log" "$OUT/step.log"
for rep in 1 2; do
  for v in 1 2; do
    echo "== VT_PSPAN=$v (round $rep)" >> "$OUT/ab.log"
    VT_PSPAN=$v timeout -k 10 200 python3 tools/bench_conv.py fwd 128,128,1,1,28 256,128,1,1,28 128,128,1,1,56 128,64,1,1,56 2>&1 | grep GF >> "$OUT/ab.log"
    VT_PSPAN=$v timeout -k 10 200 python3 tools/bench_conv.py dgrad 128,128,1,1,28 128,256,1,1,28 2>&1 | grep GF >> "$OUT/ab.log"
  done
done
cat "$OUT/ab.log"
for rep in 1 2 3; do
  for v in 1 2; do
    echo -n "[VT_PSPAN=$v] " >> "$OUT/step.log"
    VT_PSPAN=$v timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
