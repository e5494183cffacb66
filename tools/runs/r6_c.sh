#!/bin/bash
# round 6, GPU call C: kernarg placement (HIP_FORCE_DEV_KERNARG) on the step, on span6 and on its prologue stamps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6c
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_DIAG -c $CS/vt_igemm_span6.hip -o tools/diag/span6_r6d.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_r6d.so $OTHERS tools/diag/span6_r6d.o -ldl
for rep in 1 2 3; do
  for k in 0 1; do
    echo -n "[HIP_FORCE_DEV_KERNARG=$k] " >> "$OUT/step.log"
    HIP_FORCE_DEV_KERNARG=$k timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
LAYERS="128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 256,256,1,1,28 128,256,3,2,56"
for rep in 1 2; do
  for k in 0 1; do
    echo "== HIP_FORCE_DEV_KERNARG=$k (round $rep)" >> "$OUT/conv.log"
    HIP_FORCE_DEV_KERNARG=$k timeout -k 10 200 python3 tools/bench_conv.py fwd $LAYERS 2>&1 | grep GF >> "$OUT/conv.log"
  done
done
cat "$OUT/conv.log"
for k in 0 1; do
  for spec in "256 128,128,3,1,28" "256 256,256,3,1,14"; do
    set -- $spec
    echo "### HIP_FORCE_DEV_KERNARG=$k batch $1 layer $2" >> "$OUT/stamps.log"
    HIP_FORCE_DEV_KERNARG=$k VT_AMD_LIB="$ROOT/tools/diag/libvt_r6d.so" VT_SPAN6_ABL=16 VT_BENCH_BATCH=$1 timeout -k 10 120 python3 tools/bench_conv.py fwd $2 >> "$OUT/stamps.log" 2>&1
  done
done
grep -E "###|span6 stamps, us|prologue stamps" "$OUT/stamps.log" | cut -c1-330
