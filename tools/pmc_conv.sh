#!/bin/bash
# HBM bytes per launch of conv kernels on given layers (GPU box):
#   tools/pmc_conv.sh TAG [env VAR=..]... -- LAYER...        (LAYER = Cin,Cout,k,s,H as tools/bench_conv.py takes them)
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (MI355X_MICROARCH.md: one counter group per run; FETCH_SIZE is in
# KB and reports half the bytes of wide reads on gfx950 -> doubled), averaged over the launches of each kernel name.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
for e in "${ENVS[@]}"; do export "$e"; done
for grp in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_$grp" -- \
        python3 "$ROOT/tools/bench_conv.py" fwd "$@" > "$OUT/${TAG}_$grp.log" 2>&1 || exit 1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, tag = sys.argv[1:3]
acc = defaultdict(lambda: defaultdict(list))
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{tag}_{grp}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == grp and ("igemm" in r["Kernel_Name"] or "span" in r["Kernel_Name"] or "gemm6" in r["Kernel_Name"]):
            acc[(r["Kernel_Name"][:70], r["Grid_Size"], r.get("LDS_Block_Size", ""))][grp].append(float(r["Counter_Value"]))
dur = defaultdict(list)
f = glob.glob(f"{out}/{tag}_FETCH_SIZE/*/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    fe = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 1024 * 2
    wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024
    print(f"{k[0]} grid {k[1]} x{len(v['FETCH_SIZE'])}: read {fe / 1e6:8.1f} MB  write {wr / 1e6:8.1f} MB")
PY
