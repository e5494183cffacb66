"""Dev (GPU box): idle time inside a train step.  Run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 6 --warmup 3 --steps-only
then  python3 tools/trace_gaps.py <dir>: the union of kernel intervals over the last steps against wall time, the gap
histogram, and per-queue busy time (overlapped time is counted once in the union)."""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# the last ~40 % of the trace: steady-state steps
t0 = rows[int(len(rows) * 0.6)][0]
sel = [r for r in rows if r[0] >= t0]
span = sel[-1][1] - sel[0][0]
busy, cur_s, cur_e = 0, sel[0][0], sel[0][1]
gaps = []
for s, e, _, _ in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(sel)}, window {span / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms = {100 * (span - busy) / span:.1f} %")
tot = sum(e - s for s, e, _, _ in sel)
print(f"sum of kernel durations {tot / 1e6:.3f} ms -> overlapped {100 * (tot - busy) / tot:.1f} % of kernel time")
hist = defaultdict(lambda: [0, 0])
for g in gaps:
    b = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-20us" if g < 20000 else ">20us"
    hist[b][0] += 1
    hist[b][1] += g
for b in ("<1us", "1-2us", "2-4us", "4-8us", "8-20us", ">20us"):
    print(f"  gaps {b:7s}: {hist[b][0]:6d}  total {hist[b][1] / 1e6:.3f} ms")
queues = defaultdict(int)
for s, e, _, q in sel:
    queues[q] += e - s
print("  busy per queue (ms):", {q: round(v / 1e6, 2) for q, v in queues.items()})
