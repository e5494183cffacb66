#!/bin/bash
# round 6: upper bound of what folding the 134 BatchNorm finalize launches into their producers could return: the step with
# the finalize ops skipped after the first steps (coefficient rows keep real values; results are NOT those of training).
# VT_BN_FIN_APPLY=0: the program with finalize launches of its own (the default program has two of them left, NOTEBOOK R6.10)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6skipfin
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_runtime.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_DIAG_SKIP_FIN -c $CS/vt_runtime.hip -o tools/diag/rt_skipfin.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_skipfin.so $OTHERS tools/diag/rt_skipfin.o -ldl || exit 1
for rep in 1 2 3; do
  for after in 1000000000 536; do
    echo -n "[skip after $after] " >> "$OUT/step.log"
    VT_BN_FIN_APPLY=0 VT_DIAG_SKIP_FIN_AFTER=$after VT_AMD_LIB=$ROOT/tools/diag/libvt_skipfin.so timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
