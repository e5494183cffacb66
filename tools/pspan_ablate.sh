#!/bin/bash
# What bounds the persistent span kernel?  Diagnostic build (-DVT_PSPAN_DIAG) with parts switched off (wrong results by
# construction; only the time is read).  tools/pspan_ablate.sh <layer> ...   (GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/vision-toolbox_amd/csrc
mkdir -p "$ROOT/tools/diag" "$ROOT/gpurun_out"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DVT_PSPAN_DIAG -c "$CS/vt_igemm_pspan.hip" -o "$ROOT/tools/diag/pspan_diag.o" || exit 1
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_pspan.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/diag/libvt_pspandiag.so" $OTHERS "$ROOT/tools/diag/pspan_diag.o" || exit 1
for abl in ${ABLS:-0 2 4 6 8 16 24 30}; do
    echo "== VT_PSPAN_ABL=$abl (2 no stores, 4 no MFMA steps, 8 no span DMA, 16 no tile offsets)"
    VT_AMD_LIB="$ROOT/tools/diag/libvt_pspandiag.so" VT_PSPAN=2 VT_PSPAN_ABL=$abl timeout -k 10 120 python3 "$ROOT/tools/bench_conv.py" fwd "$@" 2>&1 | grep -v amdgpu.ids | grep "GF"
done
