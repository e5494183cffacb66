#!/bin/bash
# round 6: the backward BatchNorm passes (main stream) at wave priority 3 beside the filter gradients of the side stream
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6passprio
mkdir -p "$OUT"; cd "$ROOT"; rm -f "$OUT/step.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_PASS_SETPRIO=3 -c $CS/vt_elementwise.hip -o tools/diag/ew_prio.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_passprio.so $OTHERS tools/diag/ew_prio.o -ldl || exit 1
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_passprio.so"; do
    echo -n "[$(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
