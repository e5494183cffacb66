#!/bin/bash
# round 6: what each kernel family costs INSIDE the step -- the step with that family's ops skipped after the first steps
# (diagnostic build of vt_runtime.hip; the skipped ops' outputs keep the values of the warm-up steps: results are not training)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6skipkinds
mkdir -p "$OUT"; cd "$ROOT"; rm -f "$OUT/step.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_runtime.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_DIAG_SKIP_FIN -c $CS/vt_runtime.hip -o tools/diag/rt_skip.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_skip.so $OTHERS tools/diag/rt_skip.o -ldl || exit 1
run() { # label, kinds
  echo -n "[$1] " >> "$OUT/step.log"
  VT_DIAG_SKIP_KINDS="$2" VT_DIAG_SKIP_AFTER_OPS=3400 VT_AMD_LIB=$ROOT/tools/diag/libvt_skip.so timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2; do
  run "nothing skipped" ""
  run "filter gradients (conv_wgrad: the side stream)" "3"
  run "bn_bwd_reduce" "8"
  run "bn_bwd_fin_apply" "43"
  run "bn_fin_apply" "42"
  run "pointwise passes" "32,47,34,48"
  run "convolutions forward + data gradients (conv_igemm)" "2"
  run "stem backward" "29"
done
cat "$OUT/step.log"
