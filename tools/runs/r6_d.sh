#!/bin/bash
# round 6, GPU call D: why is the loader's prologue slow?  three diagnostic builds of span6 (prologue variants 0 / 1 / 2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6d
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
for v in 0 1 2; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_DIAG -DVT_S6_PV=$v -c $CS/vt_igemm_span6.hip -o tools/diag/span6_pv$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_pv$v.so $OTHERS tools/diag/span6_pv$v.o -ldl ) &
done
wait
for v in 0 1 2; do
  for spec in "256 128,128,3,1,28" "256 256,256,3,1,14"; do
    set -- $spec
    echo "### variant $v batch $1 layer $2" >> "$OUT/stamps.log"
    VT_AMD_LIB="$ROOT/tools/diag/libvt_pv$v.so" VT_SPAN6_ABL=16 VT_BENCH_BATCH=$1 timeout -k 10 120 python3 tools/bench_conv.py fwd $2 >> "$OUT/stamps.log" 2>&1
  done
done
grep -E "###|span6 stamps, us|prologue stamps" "$OUT/stamps.log" | cut -c1-360
