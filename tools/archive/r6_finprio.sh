#!/bin/bash
# round 6: the finalize arithmetic at wave priority 3 (beside the filter-gradient waves of the side stream) -- step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6finprio
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_FIN_SETPRIO=1 -c $CS/vt_elementwise.hip -o tools/diag/ew_finprio.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_finprio.so $OTHERS tools/diag/ew_finprio.o -ldl || exit 1
run() { # label, env...
  echo -n "[$1] " >> "$OUT/step.log"; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2 3; do
  run "separate launches" VT_FIN_TAIL=0
  run "separate launches, priority 3" VT_FIN_TAIL=0 VT_AMD_LIB=$ROOT/tools/diag/libvt_finprio.so
  run "bwd tails" VT_FIN_TAIL=1
  run "bwd tails, priority 3" VT_FIN_TAIL=1 VT_AMD_LIB=$ROOT/tools/diag/libvt_finprio.so
done
cat "$OUT/step.log"
