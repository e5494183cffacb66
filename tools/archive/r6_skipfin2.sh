#!/bin/bash
# round 6: where the 1.17 ms of the skipped finalize launches sits (forward / backward), and what the ticket alone costs in the step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6skipfin
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_runtime.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_DIAG_SKIP_FIN -c $CS/vt_runtime.hip -o tools/diag/rt_skipfin.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_skipfin.so $OTHERS tools/diag/rt_skipfin.o -ldl || exit 1
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_TAIL_DIAG_NOWORK -c $CS/vt_elementwise.hip -o tools/diag/ew_nowork.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_nowork.so $OTHERS tools/diag/ew_nowork.o -ldl || exit 1
run() { # label, env...
  echo -n "[$1] " >> "$OUT/step2.log"; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step2.log" || echo failed >> "$OUT/step2.log"
}
for rep in 1 2 3; do
  run "separate launches" VT_FIN_TAIL=0
  run "bwd tails" VT_FIN_TAIL=1
  run "bwd tails, ticket only after 6 steps" VT_FIN_TAIL=1 VT_DIAG_NOWORK_AFTER=342 VT_AMD_LIB=$ROOT/tools/diag/libvt_nowork.so
done
cat "$OUT/step2.log"
