#!/bin/bash
# round 6, GPU call A: graph soak, dp-graph test, hog probe (+ per-launch histograms), smoke
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6a
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -k "captured or validation or hipgraph" > "$OUT/tests.log" 2>&1
echo "tests exit $?" | tee -a "$OUT/status.txt"
tail -3 "$OUT/tests.log"
timeout -k 10 400 python tools/diag/graph_soak.py 200 darknet19 > "$OUT/soak.log" 2>&1
rc=$?; echo "soak exit $rc" | tee -a "$OUT/status.txt"; tail -5 "$OUT/soak.log"
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
rc=$?; echo "smoke exit $rc" | tee -a "$OUT/status.txt"; tail -4 "$OUT/smoke.log"
[ $rc -eq 124 ] && exit 1
for B in 128 256; do
  timeout -k 10 300 python tools/rccl_hog.py $B 20 all > "$OUT/hog_$B.log" 2>&1
  rc=$?; echo "hog $B exit $rc" | tee -a "$OUT/status.txt"; tail -2 "$OUT/hog_$B.log" | cut -c1-600
  [ $rc -eq 124 ] && exit 1
done
cd /tmp && export TMPDIR=/tmp
for m in nohog hog; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$m" -- python3 "$ROOT/tools/rccl_hog.py" 128 10 $m > "$OUT/trace_$m.log" 2>&1
  rc=$?; echo "trace $m exit $rc" | tee -a "$OUT/status.txt"
  [ $rc -eq 124 ] && exit 1
done
python3 "$ROOT/tools/hog_hist.py" "$OUT/trace_nohog" "$OUT/trace_hog" > "$OUT/hog_hist.json" 2> "$OUT/hog_hist.err"
echo "hist exit $?" | tee -a "$OUT/status.txt"
# the traces themselves are large: keep the summaries only
rm -rf "$OUT/trace_nohog" "$OUT/trace_hog"
head -c 1500 "$OUT/hog_hist.json"
