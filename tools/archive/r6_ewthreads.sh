#!/bin/bash
# round 6: 128-thread workgroups for every streaming kernel of vt_elementwise.hip (finer granularity beside the side stream)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6ewthreads
mkdir -p "$OUT"; cd "$ROOT"; rm -f "$OUT/step.log" "$OUT/tests.log"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_EW_THREADS=128 -c $CS/vt_elementwise.hip -o tools/diag/ew_128.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_ew128.so $OTHERS tools/diag/ew_128.o -ldl || exit 1
VT_AMD_LIB=$ROOT/tools/diag/libvt_ew128.so timeout -k 10 600 python -m pytest tests/test_bn_fin_apply_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a "$OUT/tests.log"
run() { # label, lib, env
  echo -n "[$1] " >> "$OUT/step.log"; lib=$2; shift; shift
  env "$@" VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
}
for rep in 1 2 3; do
  run "256 threads" "$CS/libvt_amd.so" A=1
  run "128 threads" "$ROOT/tools/diag/libvt_ew128.so" A=1
  run "128 threads, 3072 pass workgroups, 2048 reduce blocks" "$ROOT/tools/diag/libvt_ew128.so" VT_BN_FIN_APPLY_WGS=3072 VT_BN_RED_BLOCKS=2048
done
cat "$OUT/step.log"
