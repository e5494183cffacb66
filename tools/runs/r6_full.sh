#!/bin/bash
# round 6: the whole GPU suite + smoke on the final tree, as the driver runs them
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r6full}
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/ -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?; echo "gpu suite exit $rc" | tee -a "$OUT/status.txt"; tail -4 "$OUT/tests.log" | cut -c1-300
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
rc=$?; echo "smoke exit $rc" | tee -a "$OUT/status.txt"; tail -5 "$OUT/smoke.log"
