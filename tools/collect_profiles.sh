#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers rest on (run on the GPU box via gpurun):
#   tools/collect_profiles.sh <tag>          e.g. r02_a
#   1. kernel-trace + stats of the exact bench command                 -> gpurun_out/<tag>_prof_bench/
#   2. PMC passes of the dominant kernel on the three dominant 3x3 shapes, each counter group in
#      its OWN run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots):
#         FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES | GRBM_GUI_ACTIVE
#   3. JSON summaries                                                  -> gpurun_out/<tag>_*.json
# Copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof_bench" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary > "$OUT/${TAG}_prof_bench.log" 2>&1
echo "bench profile exit $?"

LAYERS="128,128,3,1,28 256,256,3,1,14 512,512,3,1,7"
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" GRBM_GUI_ACTIVE; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_pmc_$name" -- \
        python3 "$ROOT/tools/bench_conv.py" fwd $LAYERS > "$OUT/${TAG}_pmc_$name.log" 2>&1
    echo "pmc $name exit $?"
done

# 4. HBM bytes of the WHOLE step: FETCH_SIZE and WRITE_SIZE over every kernel of `bench.py --steps-only` (1 warm-up + 3
#    timed steps, nothing else launched), each counter in its own pass
for grp in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_step_pmc_$grp" -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --steps-only --no-cpu-baseline --no-pmc > "$OUT/${TAG}_step_pmc_$grp.log" 2>&1
    echo "step pmc $grp exit $?"
done
python3 - "$OUT" "$TAG" <<'EOF2'
import csv, glob, json, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
res = {"steps_counted": 4, "note": "1 warm-up + 3 timed steps of bench.py --steps-only; FETCH_SIZE x2 (gfx950 reports half the bytes of wide reads), KB -> bytes"}
per_kernel = defaultdict(lambda: [0.0, 0.0, 0])
for grp, slot in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    try:
        f = glob.glob(f"{out}/{tag}_step_pmc_{grp}/*/*counter_collection.csv")[0]
    except IndexError:
        res["error"] = f"no counter file for {grp}"
        continue
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != grp:
            continue
        v = float(r["Counter_Value"]) * 1024 * (2 if grp == "FETCH_SIZE" else 1)
        tot += v
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0].split("<")[0][-48:]
        per_kernel[k][slot] += v
        per_kernel[k][2] += 1 if slot == 0 else 0
    res[grp + "_bytes_per_step"] = tot / 4
res["hbm_bytes_per_step"] = res.get("FETCH_SIZE_bytes_per_step", 0) + res.get("WRITE_SIZE_bytes_per_step", 0)
res["by_kernel_GB_per_step"] = {k: [round(v[0] / 4e9, 3), round(v[1] / 4e9, 3), v[2] // 4] for k, v in
                                sorted(per_kernel.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:25]}
json.dump(res, open(f"{out}/{tag}_step_hbm_bytes.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "by_kernel_GB_per_step"}, indent=1))
EOF2

python3 - "$OUT" "$TAG" <<'EOF'
import csv, glob, json, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
# the three shapes are launched in this order, 25 launches each (5 warm-up + 20 timed): split by dispatch order
SHAPES = [("conv3x3 s1 128->128 @28x28 B=256", 200704, 128, 1152), ("conv3x3 s1 256->256 @14x14 B=256", 50176, 256, 2304),
          ("conv3x3 s1 512->512 @7x7 B=256", 12544, 512, 4608)]
def conv_rows(path):
    rows = [r for r in csv.DictReader(open(path)) if "span_kernel" in r["Kernel_Name"] or "span6_kernel" in r["Kernel_Name"] or "igemm_kernel" in r["Kernel_Name"]]
    return rows
def per_shape(d, counter=None):
    """{shape index: [values]} for a counter (or durations in us when counter is None)"""
    res = defaultdict(list)
    if counter is None:
        f = glob.glob(f"{out}/{tag}_pmc_{d}/*/*kernel_trace.csv")[0]
        rows = conv_rows(f)
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for i, r in enumerate(rows):
            res[i // 25].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        names = [rows[k * 25]["Kernel_Name"] for k in range(len(rows) // 25)]
        return res, names
    f = glob.glob(f"{out}/{tag}_pmc_{d}/*/*counter_collection.csv")[0]
    rows = [r for r in conv_rows(f) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for i, r in enumerate(rows):
        res[i // 25].append(float(r["Counter_Value"]))
    return res, None
avg = lambda v: sum(v) / max(len(v), 1)
summary = []
try:
    fetch, _ = per_shape("FETCH_SIZE", "FETCH_SIZE")
    write, _ = per_shape("WRITE_SIZE", "WRITE_SIZE")
    mfma, _ = per_shape("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")
    sqbusy, _ = per_shape("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")
    wavecyc, _ = per_shape("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES")
    gui, _ = per_shape("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE")
    dur, names = per_shape("GRBM_GUI_ACTIVE")
    dur_m, _ = per_shape("SQ_VALU_MFMA_BUSY_CYCLES")
    for k, (layer, M, N, K) in enumerate(SHAPES):
        flop = 2.0 * M * N * K
        n_mfma = flop / (2 * 16 * 16 * 32)
        e = {"layer": layer, "kernel": names[k] if names and k < len(names) else "?",
             "launches": len(dur[k]),
             "avg_duration_us_GRBM_pass": avg(dur[k][5:]), "avg_duration_us_SQ_pass": avg(dur_m[k][5:]),
             "flop_per_launch": flop, "mfma_16x16x32_per_launch": n_mfma,
             "FETCH_SIZE_KB_raw": avg(fetch[k]), "WRITE_SIZE_KB": avg(write[k]),
             # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled
             "hbm_traffic_bytes": avg(fetch[k]) * 1024 * 2 + avg(write[k]) * 1024,
             "algorithmic_bytes": 2.0 * (M * K / 9 + N * K + M * N),
             "SQ_VALU_MFMA_BUSY_CYCLES": avg(mfma[k]), "SQ_BUSY_CYCLES": avg(sqbusy[k]), "SQ_WAVE_CYCLES": avg(wavecyc[k]),
             "GRBM_GUI_ACTIVE_sum_over_8_XCDs": avg(gui[k])}
        # issue-slot cycles the MFMAs of one launch need at 16 cycles each, spread over 1024 SIMDs
        e["mfma_busy_cycles_per_mfma"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / n_mfma
        e["kernel_cycles"] = e["GRBM_GUI_ACTIVE_sum_over_8_XCDs"] / 8
        e["effective_clock_GHz"] = e["kernel_cycles"] / (e["avg_duration_us_GRBM_pass"] * 1e3)
        e["mfma_pipe_utilisation"] = (n_mfma * 16 / 1024) / e["kernel_cycles"]
        e["tflops_SQ_pass"] = flop / e["avg_duration_us_SQ_pass"] / 1e6
        summary.append(e)
except Exception as ex:  # keep whatever was collected
    summary.append({"error": repr(ex)})
json.dump(summary, open(f"{out}/{tag}_dominant_kernel_pmc.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
EOF

# 5. (round 4) HBM bytes of the persistent span kernel on its layers: 32->32 3x3 @112x112 and 32->64 3x3 stride 2 @224x224
for grp in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_pspan_pmc_$grp" -- \
        python3 "$ROOT/tools/bench_conv.py" fwd 32,32,3,1,112 32,64,3,2,224 64,64,3,1,56 > "$OUT/${TAG}_pspan_pmc_$grp.log" 2>&1
    echo "pspan pmc $grp exit $?"
done
python3 - "$OUT" "$TAG" <<'EOF3'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
LAY = [("conv3x3 s1 32->32 @112x112 B=256", 256 * 112 * 112 * (32 + 32) * 2 + 32 * 288 * 2),
       ("conv3x3 s2 32->64 @224x224 B=256", 256 * (224 * 224 * 32 + 112 * 112 * 64) * 2 + 64 * 288 * 2),
       ("conv3x3 s1 64->64 @56x56 B=256", 256 * 56 * 56 * (64 + 64) * 2 + 64 * 576 * 2)]
res = []
try:
    vals = {}
    for grp in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/{tag}_pspan_pmc_{grp}/*/*counter_collection.csv")[0]
        rows = [r for r in csv.DictReader(open(f)) if "pspan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == grp]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals[grp] = [float(r["Counter_Value"]) for r in rows]
    f = glob.glob(f"{out}/{tag}_pspan_pmc_FETCH_SIZE/*/*kernel_trace.csv")[0]
    tr = [r for r in csv.DictReader(open(f)) if "pspan_kernel" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(tr) // len(LAY)
    for k, (name, alg) in enumerate(LAY):
        sl = slice(k * n + 5, (k + 1) * n)
        fe = sum(vals["FETCH_SIZE"][sl]) / max(1, len(vals["FETCH_SIZE"][sl]))
        wr = sum(vals["WRITE_SIZE"][sl]) / max(1, len(vals["WRITE_SIZE"][sl]))
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr[sl]]
        res.append({"layer": name, "kernel": tr[k * n]["Kernel_Name"][:80], "launches": len(du), "avg_duration_us_pmc_pass": sum(du) / len(du),
                    "FETCH_SIZE_KB_raw": fe, "WRITE_SIZE_KB": wr, "hbm_traffic_bytes": fe * 2048 + wr * 1024,
                    "algorithmic_bytes": alg, "traffic_over_algorithmic": (fe * 2048 + wr * 1024) / alg})
except Exception as ex:
    res.append({"error": repr(ex)})
json.dump(res, open(f"{out}/{tag}_pspan_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
EOF3

# 6. (round 5) the CU-owning filter-gradient kernel on the dominant shape: 25 launches of one layer, then 25 launches of a
#    group of eight layers (tools/bench_conv.py wgrad with VT_BENCH_GROUP=8); each counter group in its own pass
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" GRBM_GUI_ACTIVE; do
    name=$(echo $grp | cut -d' ' -f1)
    VT_BENCH_GROUP=8 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_w6_pmc_$name" -- \
        python3 "$ROOT/tools/bench_conv.py" wgrad 128,128,3,1,28 > "$OUT/${TAG}_w6_pmc_$name.log" 2>&1
    echo "wgrad6 pmc $name exit $?"
done
python3 - "$OUT" "$TAG" <<'EOF6'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"layer": "filter gradient of conv3x3 s1 128->128 @28x28 B=256 (59.19 GFLOP, 103 MB of operands per layer)",
       "note": "dispatch order: 25 launches of ONE layer (split 64), then 25 launches of a GROUP OF EIGHT layers (split 8 each); FETCH_SIZE x2 on gfx950"}
def rows(d, counter):
    f = glob.glob(f"{out}/{tag}_w6_pmc_{d}/*/*counter_collection.csv")[0]
    r = [x for x in csv.DictReader(open(f)) if "wgrad6_kernel" in x["Kernel_Name"] and x["Counter_Name"] == counter]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return [float(x["Counter_Value"]) for x in r]
def durs(d):
    f = glob.glob(f"{out}/{tag}_w6_pmc_{d}/*/*kernel_trace.csv")[0]
    r = [x for x in csv.DictReader(open(f)) if "wgrad6_kernel" in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in r]
avg = lambda v: sum(v) / max(len(v), 1)
try:
    for label, lo, hi, layers in (("one_layer", 5, 25, 1), ("group_of_8", 30, 50, 8)):
        fe, wr = rows("FETCH_SIZE", "FETCH_SIZE")[lo:hi], rows("WRITE_SIZE", "WRITE_SIZE")[lo:hi]
        mf = rows("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")[lo:hi]
        gui = rows("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE")[lo:hi]
        du = durs("GRBM_GUI_ACTIVE")[lo:hi]
        flop = 59190018048.0 * layers
        n_mfma = flop * (29 * 29) / (28 * 28) / (2 * 16 * 16 * 32)  # (padded positions: 29 x 29 per image)
        e = {"launches": len(du), "layers_per_launch": layers, "avg_duration_us": avg(du), "us_per_layer": avg(du) / layers,
             "tflops": flop / avg(du) / 1e6, "frac_of_2.5PF": flop / avg(du) / 1e6 / 2500.0,
             "hbm_traffic_bytes_per_layer": (avg(fe) * 2048 + avg(wr) * 1024) / layers,
             "FETCH_SIZE_KB_raw": avg(fe), "WRITE_SIZE_KB": avg(wr), "SQ_VALU_MFMA_BUSY_CYCLES": avg(mf),
             "kernel_cycles": avg(gui) / 8, "effective_clock_GHz": avg(gui) / 8 / (avg(du) * 1e3),
             "mfma_pipe_utilisation": (n_mfma * 16 / 1024) / (avg(gui) / 8)}
        res[label] = e
except Exception as ex:
    res["error"] = repr(ex)
json.dump(res, open(f"{out}/{tag}_wgrad6_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
EOF6
