#!/bin/bash
# round 6: isolated cost of the finalize tail (ticket alone / ticket + work) against the finalize launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6fintail
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_elementwise.o)
/opt/rocm/bin/hipcc $FLAGS -DVT_TAIL_DIAG_NOWORK -c $CS/vt_elementwise.hip -o tools/diag/ew_nowork.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_nowork.so $OTHERS tools/diag/ew_nowork.o -ldl || exit 1
echo "== shipped" | tee -a "$OUT/iso.log"
timeout -k 10 300 python3 tools/bench_fin_tail.py 2>&1 | grep "^M" | tee -a "$OUT/iso.log"
echo "== ticket only (diagnostic)" | tee -a "$OUT/iso.log"
VT_AMD_LIB=$ROOT/tools/diag/libvt_nowork.so timeout -k 10 300 python3 tools/bench_fin_tail.py 2>&1 | grep "^M" | tee -a "$OUT/iso.log"
