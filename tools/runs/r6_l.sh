#!/bin/bash
# round 6, GPU call L: necks with fuse_fn="concat", span6 with s_setprio in the MFMA tick
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6l
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_necks.py tests/test_span6_gpu.py tests/test_dgrad_bnred_gpu.py -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?; echo "tests exit $rc" | tee -a "$OUT/status.txt"; tail -12 "$OUT/tests.log" | cut -c1-250
for rep in 1 2; do
  echo -n "[default] " >> "$OUT/step.log"
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
done
cat "$OUT/step.log"
