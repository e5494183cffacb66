#!/bin/bash
# round 6: workgroups of 512 / 1024 threads for the self-finalizing passes (fewer workgroups re-read the statistics) x grid cuts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6finthreads
mkdir -p "$OUT"; cd "$ROOT"; rm -f "$OUT/step.log" "$OUT/tests.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
for t in 512 1024; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVT_FIN_THREADS=$t -c $CS/vt_elementwise.hip -o tools/diag/ew_t$t.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_t$t.so $OTHERS tools/diag/ew_t$t.o -ldl ) &
done
wait
for t in 512 1024; do
  VT_AMD_LIB=$ROOT/tools/diag/libvt_t$t.so timeout -k 10 300 python -m pytest tests/test_bn_fin_apply_gpu.py -x -q -m gpu 2>&1 | tail -1 | tee -a "$OUT/tests.log"
done
run() { # label, lib, wgs
  echo -n "[$1] " >> "$OUT/step.log"
  VT_BN_FIN_APPLY_WGS=$3 VT_AMD_LIB="$2" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2; do
  run "256 threads x 1536" "$CS/libvt_amd.so" 1536
  run "512 threads x 768" "$ROOT/tools/diag/libvt_t512.so" 768
  run "512 threads x 1536" "$ROOT/tools/diag/libvt_t512.so" 1536
  run "1024 threads x 384" "$ROOT/tools/diag/libvt_t1024.so" 384
  run "1024 threads x 768" "$ROOT/tools/diag/libvt_t1024.so" 768
done
cat "$OUT/step.log"
