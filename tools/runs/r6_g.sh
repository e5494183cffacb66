#!/bin/bash
# round 6, GPU call G: fused dgrad + BatchNorm-backward sums -- parity, isolated timing, step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r6g}
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_span6_gpu.py tests/test_dgrad_bnred_gpu.py -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?; echo "kernel tests exit $rc" | tee -a "$OUT/status.txt"; tail -3 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
  timeout -k 10 200 python3 tools/bench_conv.py bnred 128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 2>&1 | grep GF >> "$OUT/bnred.log"
done
cat "$OUT/bnred.log"
for rep in 1 2 3; do
  for cfg in "VT_FUSE_BNRED=0" "VT_FUSE_BNRED=1"; do
    echo -n "[$cfg] " >> "$OUT/step.log"
    env $cfg timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
