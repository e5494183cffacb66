#!/bin/bash
# round 6: pixel rows (masked) against padded positions for the dominant 128 -> 128 @28x28 layer, in situ and alone
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6mask
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  for m in 1 2; do
    echo "== VT_SPAN6_MASK=$m (round $rep)" >> "$OUT/conv.log"
    VT_SPAN6_MASK=$m timeout -k 10 200 python3 tools/bench_conv.py fwd 128,128,3,1,28 160,160,3,1,28 128,128,3,1,56 2>&1 | grep GF >> "$OUT/conv.log"
    VT_BENCH_RESIDUAL=1 VT_SPAN6_MASK=$m timeout -k 10 200 python3 tools/bench_conv.py fwd 128,128,3,1,28 2>&1 | grep GF | sed 's/^/[+res] /' >> "$OUT/conv.log"
  done
done
cat "$OUT/conv.log"
for rep in 1 2; do
  for m in 1 2; do
    VT_SPAN6_MASK=$m timeout -k 10 400 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > "$OUT/bench_m${m}_$rep.json" 2>/dev/null
    python3 - "$OUT/bench_m${m}_$rep.json" $m <<'EOF'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("MASK", sys.argv[2], "step", d["ms_per_step"], "frac", r["frac"], "launch_ms", r["launch_ms"], "rocprof", r.get("launch_ms_rocprof"), r.get("frac_rocprof"), r["kernel"])
l=d["roofline_layers"][0]; print("   ", {k:(v["frac"],v["ms"]) for k,v in l.items() if isinstance(v,dict)})
EOF
  done
done
