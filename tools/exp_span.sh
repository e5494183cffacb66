#!/bin/bash
# GPU box: ablation timings of the span conv kernel (diagnostic builds from tools/build_diag.sh)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
L="128,128,3,1,28 256,256,3,1,14 128,128,3,1,56 256,256,3,1,28"
echo "== shipped"; python3 tools/bench_conv.py fwd $L
for k in 1 2 3 4 5 6; do
  echo "== ablate $k (1 no MFMA, 2 no DMA in loop, 4 no fragment reads)"
  VT_AMD_LIB=$ROOT/tools/diag/libvt_abl$k.so python3 tools/bench_conv.py fwd $L
done
echo "== stamps"; VT_AMD_LIB=$ROOT/tools/diag/libvt_stamps.so python3 tools/bench_conv.py fwd 128,128,3,1,28 2>&1
for v in "VT_SPAN_BM=256" "VT_SPAN_BM=512" "VT_SPAN_BM=512 VT_SPAN_PP=0" "VT_SPAN_BM=256 VT_SPAN_WAVES=8" "VT_SPAN_PD=3"; do
  echo "== $v"; env $v python3 tools/bench_conv.py fwd $L
done
