#!/bin/bash
# round 6: s_setprio around wgrad6's MFMA tick -- parity + alone + step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6w6prio
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_wgrad6.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_W6_SETPRIO=1 -c $CS/vt_wgrad6.hip -o tools/diag/w6_prio.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_w6prio.so $OTHERS tools/diag/w6_prio.o -ldl || exit 1
VT_AMD_LIB=$ROOT/tools/diag/libvt_w6prio.so python -m pytest tests/test_wgrad6_gpu.py -x -q -m gpu 2>&1 | tail -1 | tee -a "$OUT/tests.log"
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_w6prio.so"; do
    echo "== $(basename $lib) (round $rep)" >> "$OUT/ab.log"
    VT_BENCH_GROUP=8 VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py wgrad 128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 128,128,3,1,56 2>&1 | grep GF | sed 's/.*group of/group of/' >> "$OUT/ab.log"
  done
done
cat "$OUT/ab.log"
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_w6prio.so"; do
    echo -n "[$(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
    echo -n "[vovnet39 $(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --model vovnet39 --steps 20 --warmup 6 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
