#!/bin/bash
# round 6: does the finalize-tail code at the end of span6_kernel<1,...> (off in the engine) cost the statistics launches anything?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6s6tail
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/ab.log" "$OUT/step.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_NO_FIN_TAIL -c $CS/vt_igemm_span6.hip -o tools/diag/span6_notail.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_notail.so $OTHERS tools/diag/span6_notail.o -ldl || exit 1
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_notail.so"; do
    echo "== $(basename $lib) (round $rep)" >> "$OUT/ab.log"
    VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py fwd 128,128,3,1,28 256,256,3,1,14 2>&1 | grep GF >> "$OUT/ab.log"
  done
done
cat "$OUT/ab.log"
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_notail.so"; do
    echo -n "[$(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
