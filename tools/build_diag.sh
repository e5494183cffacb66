#!/bin/bash
# Diagnostic builds of libvt_amd (never shipped, never loaded by the package unless VT_AMD_LIB points at them):
#   tools/diag/libvt_abl<k>.so  = span kernel built with -DVT_SPAN_ABLATE=<k>  (k in $ABLS)
#   tools/diag/libvt_stamps.so  = span kernel built with -DVT_SPAN_STAMPS
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/vision-toolbox_amd/csrc
OUT=$ROOT/tools/diag
mkdir -p "$OUT"
make -C "$CS" -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span.o)
for k in ${ABLS:-1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN_ABLATE=$k -c "$CS/vt_igemm_span.hip" -o "$OUT/span_abl$k.o" &
done
/opt/rocm/bin/hipcc $FLAGS -DVT_SPAN_STAMPS -c "$CS/vt_igemm_span.hip" -o "$OUT/span_stamps.o" &
wait
for k in ${ABLS:-1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvt_abl$k.so" $OTHERS "$OUT/span_abl$k.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvt_stamps.so" $OTHERS "$OUT/span_stamps.o"
# span3 ablations
OTHERS3=$(ls "$CS"/*.o | grep -v vt_igemm_span3.o)
for k in ${ABLS3:-1 2 3 4 8 15}; do
  /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN3_ABLATE=$k -c "$CS/vt_igemm_span3.hip" -o "$OUT/span3_abl$k.o" &
done
wait
for k in ${ABLS3:-1 2 3 4 8 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvt_s3abl$k.so" $OTHERS3 "$OUT/span3_abl$k.o"
done
ls -la "$OUT"/*.so
