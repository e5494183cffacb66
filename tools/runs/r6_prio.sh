#!/bin/bash
# round 6: s_setprio in span6 (1: around the MFMA tick, 2: loaders raised, 3: both) -- bit-equality + step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6prio
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
for v in 1 2 3; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_SETPRIO=$v -c $CS/vt_igemm_span6.hip -o tools/diag/span6_prio$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_prio$v.so $OTHERS tools/diag/span6_prio$v.o -ldl ) &
done
wait
for v in 1 2 3; do
  VT_AMD_LIB=$ROOT/tools/diag/libvt_prio$v.so python -m pytest tests/test_span6_gpu.py -x -q -m gpu 2>&1 | tail -1 | tee -a "$OUT/tests.log"
done
LAYERS="128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 128,128,3,1,56"
for rep in 1 2; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_prio1.so" "$ROOT/tools/diag/libvt_prio2.so" "$ROOT/tools/diag/libvt_prio3.so"; do
    echo "== $(basename $lib) (round $rep)" >> "$OUT/ab.log"
    VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py fwd $LAYERS 2>&1 | grep GF >> "$OUT/ab.log"
  done
done
cat "$OUT/ab.log"
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_prio1.so" "$ROOT/tools/diag/libvt_prio2.so" "$ROOT/tools/diag/libvt_prio3.so"; do
    echo -n "[$(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
