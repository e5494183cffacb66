#!/bin/bash
# round 6: the evidence bench.py's numbers rest on, collected on the FINAL tree (VERDICT r05 #7a) -> gpurun_out/<tag>_*
#   1. tools/collect_profiles.sh <tag>: rocprofv3 --kernel-trace --stats of the bench command, PMC passes of the dominant kernel,
#      HBM bytes of the whole step, pspan PMC
#   2. the in-situ kernel trace of the step (tools/trace_insitu.py) and the isolated per-op profile (tools/profile_ops.py)
#   3. span6's phase stamps (tools/span6_phases.sh)
#   4. the full default bench line (what the driver runs)
set -u
TAG=${1:-r06_b}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
bash tools/collect_profiles.sh "$TAG" > "$OUT/${TAG}_collect.log" 2>&1
echo "collect exit $?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_insitu_trace" -- \
    python3 "$ROOT/bench.py" --steps 6 --warmup 3 --steps-only --no-cpu-baseline --no-pmc --no-secondary > "$OUT/${TAG}_insitu_trace.log" 2>&1 )
python3 tools/trace_insitu.py "$OUT/${TAG}_insitu_trace" 4 > "$OUT/${TAG}_insitu.txt" 2>&1
echo "insitu exit $?"; head -30 "$OUT/${TAG}_insitu.txt"
rm -rf "$OUT/${TAG}_insitu_trace"
timeout -k 10 400 python3 tools/profile_ops.py cspdarknet53 256 60 > "$OUT/${TAG}_ops_profile_cspdarknet53.txt" 2>&1
echo "ops profile exit $?"; head -28 "$OUT/${TAG}_ops_profile_cspdarknet53.txt"
timeout -k 10 400 python3 tools/profile_ops.py vovnet39 256 40 > "$OUT/${TAG}_ops_profile_vovnet39.txt" 2>&1
echo "ops profile vovnet exit $?"
bash tools/span6_phases.sh "$TAG" > "$OUT/${TAG}_phases.log" 2>&1
echo "phases exit $?"
timeout -k 10 900 python3 bench.py > "$OUT/${TAG}_bench_line.json" 2> "$OUT/${TAG}_bench_stderr.log"
echo "bench exit $?"; cut -c1-1500 "$OUT/${TAG}_bench_line.json"
