#!/bin/bash
# A/B of two builds of libvt_amd on conv layers, alternating runs on ONE box:  tools/ab_conv.sh <libA> <libB> [env...] -- layers...
A=$1; B=$2; shift 2
ENVV=()
while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    echo "== $lib (round $rep)"
    env "${ENVV[@]}" VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py fwd "$@" 2>&1 | grep GF
  done
done
