#!/bin/bash
# round 6: every workgroup of the streaming pass finalizes its own channel group (vt_bn_finalize_apply / vt_bn_bwd_finalize_apply,
# second form) -- parity + step A/B against the separate finalize launches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6selffin
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/tests.log" "$OUT/step.log"
timeout -k 10 600 python -m pytest tests/test_bn_fin_apply_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a "$OUT/tests.log"
grep -q passed "$OUT/tests.log" && ! grep -q failed "$OUT/tests.log" || exit 1
run() { # label, env...
  echo -n "[$1] " >> "$OUT/step.log"; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2 3; do
  run "separate launches" VT_FIN_TAIL=0 VT_BN_FIN_APPLY=0
  run "self-finalizing passes, 1024 workgroups" VT_FIN_TAIL=0 VT_BN_FIN_APPLY=1
  run "self-finalizing passes, 512 workgroups" VT_FIN_TAIL=0 VT_BN_FIN_APPLY=1 VT_BN_FIN_APPLY_WGS=512
  run "self-finalizing passes, 2048 workgroups" VT_FIN_TAIL=0 VT_BN_FIN_APPLY=1 VT_BN_FIN_APPLY_WGS=2048
done
cat "$OUT/step.log"
