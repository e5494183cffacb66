"""Per-launch duration histograms of the CU-owning kernels with and without a resident hog (VERDICT r05 #5d).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d D0 -- python3 $ROOT/tools/rccl_hog.py 128 10 nohog
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d D1 -- python3 $ROOT/tools/rccl_hog.py 128 10 hog
    python3 tools/hog_hist.py D0 D1 > profiles/r06_cu_hog_launch_hist.json

For every kernel family of interest: launches, median / p90 / max duration, and the share of launches longer than 1.5x
the no-hog median of the SAME family (a CU-owning launch that fell into two rounds takes ~2x).  For the hog run also: how
many launches of the family STARTED while a hog workgroup was resident, and those launches' statistics alone."""
import csv
import glob
import json
import re
import statistics
import sys
from collections import defaultdict

FAMS = ("span6_kernel", "wgrad6_kernel", "pspan_kernel", "span_kernel", "wgrad_kernel", "igemm_kernel", "pw_kernel",
        "bn_bwd_reduce_kernel", "bn_bwd_apply_kernel")


def load(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def fam(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"([A-Za-z0-9_]+_kernel)", k)
    return m.group(1) if m else None


def stats(v):
    v = sorted(v)
    return {"n": len(v), "median_us": round(statistics.median(v) / 1e3, 2), "p90_us": round(v[int(0.9 * (len(v) - 1))] / 1e3, 2),
            "max_us": round(v[-1] / 1e3, 2)} if v else {"n": 0}


def main():
    base, hog = load(sys.argv[1]), load(sys.argv[2])
    out = {"no_hog": {}, "hog": {}, "note": "durations of every launch in the trace (warm-up included); shape mix is the same in both runs"}
    med = {}
    by = defaultdict(list)
    for s, e, k in base:
        f = fam(k)
        if f in FAMS:
            by[f].append(e - s)
    for f, v in by.items():
        out["no_hog"][f] = stats(v)
        med[f] = statistics.median(v)
    hogs = [(s, e) for s, e, k in hog if "hog_kernel" in k and e - s > 50_000]  # (the W-wide launches: >= 50 us)
    out["hog"]["hog_launches"] = len(hogs)
    out["hog"]["hog_resident_ms_total"] = round(sum(e - s for s, e in hogs) / 1e6, 3)
    by = defaultdict(list)
    inside = defaultdict(list)
    for s, e, k in hog:
        f = fam(k)
        if f in FAMS:
            by[f].append(e - s)
            if any(hs <= s < he for hs, he in hogs):
                inside[f].append(e - s)
    for f, v in by.items():
        st = stats(v)
        if f in med:
            st["share_longer_than_1p5x_no_hog_median"] = round(sum(1 for x in v if x > 1.5 * med[f]) / len(v), 4)
        st["started_while_hog_resident"] = stats(inside[f])
        if inside[f] and f in med:
            st["started_while_hog_resident"]["share_longer_than_1p5x_no_hog_median"] = round(
                sum(1 for x in inside[f] if x > 1.5 * med[f]) / len(inside[f]), 4)
        out["hog"][f] = st
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
