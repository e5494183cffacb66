#!/bin/bash
# GPU box: span3 kernel timings, with ablation builds
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
L="${LAYERS:-128,128,3,1,28 256,256,3,1,14 128,128,3,1,56}"
export VT_SPAN3=1
for wm in 4 2; do
  export VT_SPAN3_WM=$wm
  echo "== span3 WM=$wm"; python3 tools/bench_conv.py fwd $L 2>&1 | grep -v amdgpu.ids
  for k in ${ABLS3:-1 2 3 4 8 15}; do
    [ -f tools/diag/libvt_s3abl$k.so ] || continue
    echo "== span3 WM=$wm ablate $k (1 no MFMA, 2 no DMA in loop, 4 no fragment reads, 8 no stores)"
    VT_AMD_LIB=$ROOT/tools/diag/libvt_s3abl$k.so python3 tools/bench_conv.py fwd $L 2>&1 | grep -v amdgpu.ids
  done
done
