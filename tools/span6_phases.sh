#!/bin/bash
# Phase breakdown of the dominant kernel (span6) from its diagnostic stamps -> gpurun_out/<tag>_span6_phases.json
#   tools/span6_phases.sh <tag>      (GPU box; builds tools/diag/libvt_span6diag.so with -DVT_SPAN6_DIAG first)
# The shipped library carries no stamp code; the diagnostic build is loaded through VT_AMD_LIB for these runs only.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/vision-toolbox_amd/csrc
OUT=$ROOT/gpurun_out
mkdir -p "$OUT" "$ROOT/tools/diag"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_DIAG -c "$CS/vt_igemm_span6.hip" -o "$ROOT/tools/diag/span6_diag.o" || exit 1
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/diag/libvt_span6diag.so" $OTHERS "$ROOT/tools/diag/span6_diag.o" || exit 1
: > "$OUT/${TAG}_span6_stamps.log"
for spec in "256 128,128,3,1,28" "256 256,256,3,1,14" "256 128,128,3,1,56" "128 128,128,3,1,28" "64 320,320,3,1,40"; do
    set -- $spec
    echo "### batch $1 layer $2" >> "$OUT/${TAG}_span6_stamps.log"
    VT_AMD_LIB="$ROOT/tools/diag/libvt_span6diag.so" VT_SPAN6_ABL=16 VT_BENCH_BATCH=$1 timeout -k 10 120 \
        python3 "$ROOT/tools/bench_conv.py" fwd $2 >> "$OUT/${TAG}_span6_stamps.log" 2>&1
done
python3 - "$OUT/${TAG}_span6_stamps.log" "$OUT/${TAG}_span6_phases.json" <<'EOF2'
import json, re, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.match(r"### batch (\d+) layer (\S+)", line)
    if m:
        cur = {"batch": int(m.group(1)), "layer": m.group(2)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    m = re.search(r"loader start ([\d.]+) prologue done ([\d.]+) loop done ([\d.]+) \| tile0: first tick ([\d.]+) loop end ([\d.]+) "
                  r"epilogue end ([\d.]+) \| tile1: ([\d.]+) ([\d.]+) ([\d.]+) \| tile2: ([\d.]+) ([\d.]+) ([\d.]+)", line)
    if m:
        v = [float(x) for x in m.groups()]
        cur["us_from_first_workgroup_start"] = {
            "loader_start": v[0], "prologue_issued": v[1], "loaders_done": v[2],
            "tile0": {"first_tick": v[3], "loop_end": v[4], "epilogue_end": v[5]},
            "tile1": {"first_tick": v[6], "loop_end": v[7], "epilogue_end": v[8]}}
        cur["phases_us"] = {"until_first_tick": v[3], "tile0_loop": round(v[4] - v[3], 2), "tile0_epilogue": round(v[5] - v[4], 2),
                            "tile1_loop": round(v[7] - v[6], 2), "tile1_epilogue": round(v[8] - v[7], 2)}
    m = re.search(r"loader 0 waiting in barriers (\d+) of (\d+) in its loop; compute wave 0 waiting in barriers (\d+), "
                  r"read-tick work (\d+), MFMA-tick work (\d+)", line)
    if m:
        cur["shader_cycles_per_workgroup"] = dict(zip(("loader0_barrier_wait", "loader0_loop", "compute0_barrier_wait",
                                                       "compute0_read_tick_work", "compute0_mfma_tick_work"), map(int, m.groups())))
    m = re.search(r"fwd\s+([\d.]+) ms\s+([\d.]+) TF/s \[(.*)\]", line)
    if m:
        cur["launch_us_with_stamps"] = float(m.group(1)) * 1e3
        cur["kernel"] = m.group(3)
json.dump({"note": "vt_igemm_span6.hip built with -DVT_SPAN6_DIAG, VT_SPAN6_ABL=16: wall-clock stamps (100 MHz) per workgroup, "
                   "means over the 256 workgroups of a warm launch; the stamps themselves cost ~10 % (launch_us_with_stamps is "
                   "not the shipped kernel's time)", "shapes": rows}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
EOF2
