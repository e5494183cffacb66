#!/bin/bash
# round 6: row blocks of the BatchNorm backward reduction (VT_BN_RED_BLOCKS; 256 since round 2) and workgroups of the
# self-finalizing passes (VT_BN_FIN_APPLY_WGS) -- step sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6redblocks
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/step.log"
run() { # label, env...
  echo -n "[$1] " >> "$OUT/step.log"; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2; do
  for b in 256 128 512 1024 2048; do run "reduce blocks $b" VT_BN_RED_BLOCKS=$b; done
  for w in 1024 1536 3072 4096; do run "pass workgroups $w" VT_BN_FIN_APPLY_WGS=$w; done
done
cat "$OUT/step.log"
