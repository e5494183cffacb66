#!/bin/bash
# A/B of environment settings on the train step, alternating runs on ONE box:
#   tools/ab_env.sh "VT_A=1" "VT_A=0 VT_B=2" ... [-- bench args]      (each argument: one configuration's environment)
CFGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFGS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2 3; do
  for cfg in "${CFGS[@]}"; do
    echo -n "[$cfg] "
    env $cfg timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only "$@" 2>&1 | grep -o '"ms_per_step": [0-9.]*' || echo failed
  done
done
