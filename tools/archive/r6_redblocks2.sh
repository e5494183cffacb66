#!/bin/bash
# round 6: (reduce blocks, pass workgroups) = (256, 1024) against (1024, 1536), four alternating rounds, both models
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6redblocks
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/step2.log"
run() { # label, model, env...
  echo -n "[$1] " >> "$OUT/step2.log"; model=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --model $model --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step2.log" || echo failed >> "$OUT/step2.log"
}
for rep in 1 2 3 4; do
  run "256 / 1024" cspdarknet53 VT_BN_RED_BLOCKS=256 VT_BN_FIN_APPLY_WGS=1024
  run "1024 / 1536" cspdarknet53 VT_BN_RED_BLOCKS=1024 VT_BN_FIN_APPLY_WGS=1536
  run "vovnet39 256 / 1024" vovnet39 VT_BN_RED_BLOCKS=256 VT_BN_FIN_APPLY_WGS=1024
  run "vovnet39 1024 / 1536" vovnet39 VT_BN_RED_BLOCKS=1024 VT_BN_FIN_APPLY_WGS=1536
done
cat "$OUT/step2.log"
