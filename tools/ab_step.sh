#!/bin/bash
# A/B of two builds of libvt_amd on the train step, alternating runs on ONE box:  tools/ab_step.sh <libA> <libB> [bench args...]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    echo -n "$lib: "
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only "$@" 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  done
done
