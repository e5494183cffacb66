#!/bin/bash
# round 6: the self-finalizing passes with / without their first rows loaded in front of the finalize prologue -- step A/B
# (tools/diag/ew_noprefetch.hip: vt_elementwise.hip of the commit before)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6prefetch
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/step.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -c tools/diag/ew_noprefetch.hip -o tools/diag/ew_noprefetch.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_noprefetch.so $OTHERS tools/diag/ew_noprefetch.o -ldl || exit 1
timeout -k 10 600 python -m pytest tests/test_bn_fin_apply_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_noprefetch.so"; do
    for model in cspdarknet53 vovnet39; do
      echo -n "[$model $(basename $lib)] " >> "$OUT/step.log"
      VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --model $model --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
    done
  done
done
cat "$OUT/step.log"
