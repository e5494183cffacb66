#!/bin/bash
# Consumer-side normalise in span6's loader waves: build the prototype library here (no GPU needed),
#   tools/s6proto.sh build
# then on the GPU box time it against the shipped kernel, check conv(relu(x)) bit for bit and collect the SQ / LDS counters:
#   tools/s6proto.sh run
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/vision-toolbox_amd/csrc
OUT=$ROOT/tools/diag
mkdir -p "$OUT" "$ROOT/gpurun_out"
if [ "${1:-run}" = build ]; then
    make -C "$CS" >/dev/null || exit 1
    FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function"
    /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_PROTO_NORM -c "$CS/vt_igemm_span6.hip" -o "$OUT/s6proto.o" || exit 1
    OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvt_s6proto.so" $OTHERS "$OUT/s6proto.o" -ldl || exit 1
    rm -f "$OUT/s6proto.o"
    exit 0
fi
S="128,128,28 256,256,14 128,128,56"
for r in 1 2; do
    python3 "$ROOT/tools/s6proto_check.py" ref $S 2>&1 | grep -v amdgpu.ids
    VT_AMD_LIB="$OUT/libvt_s6proto.so" python3 "$ROOT/tools/s6proto_check.py" proto $S 2>&1 | grep -v amdgpu.ids
done
python3 "$ROOT/tools/s6proto_check.py" compare $S
bash "$ROOT/tools/pmc_s6proto.sh"
