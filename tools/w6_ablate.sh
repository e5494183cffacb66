#!/bin/bash
# What bounds the CU-owning filter-gradient kernel (vt_wgrad6.hip)?  Builds with parts compiled out (-DVT_W6_ABL=<bits>:
# wrong results by construction; only the time is read).  Build here (no GPU needed): tools/w6_ablate.sh build ; on the GPU
# box:  VT_BENCH_GROUP=8 tools/w6_ablate.sh run <layer> ...
#   bits: 1 no LDS-DMA inside the loop, 2 no MFMAs, 4 no fragment reads, 8 no flush
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/vision-toolbox_amd/csrc
OUT=$ROOT/tools/diag
ABLS=${ABLS:-0 8 9 10 12 14 15}
mkdir -p "$OUT" "$ROOT/gpurun_out"
if [ "$1" = build ]; then
    FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function"
    OTHERS=$(ls "$CS"/*.o | grep -v vt_wgrad6.o)
    for k in $ABLS; do /opt/rocm/bin/hipcc $FLAGS -DVT_W6_ABL=$k -c "$CS/vt_wgrad6.hip" -o "$OUT/w6_abl$k.o" & done
    wait
    for k in $ABLS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvt_w6abl$k.so" $OTHERS "$OUT/w6_abl$k.o" -ldl || exit 1; done
    rm -f "$OUT"/w6_abl*.o
    exit 0
fi
shift
for k in $ABLS; do
    echo "== VT_W6_ABL=$k (1 no DMA in the loop, 2 no MFMAs, 4 no fragment reads, 8 no flush)"
    VT_AMD_LIB="$OUT/libvt_w6abl$k.so" timeout -k 10 120 python3 "$ROOT/tools/bench_conv.py" wgrad "$@" 2>&1 | grep -v amdgpu.ids | grep "wgrad"
done
