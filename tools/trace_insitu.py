"""Dev (GPU box): where a train step's time goes IN SITU.  Run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 6 --warmup 3 --steps-only
then  python3 tools/trace_insitu.py <dir> [steps=4]:
  * per kernel family: launches and summed duration per step (to set against tools/profile_ops.py's isolated times),
  * the main queue's idle time inside a step (gaps between its consecutive kernels) by size class,
  * how much of the side queue's busy time overlaps main-queue kernels."""
import csv
import glob
import re
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# a step ends with the SGD launches; the loss kernel runs once per step
marks = [i for i, r in enumerate(rows) if "xent" in r[2]]
assert len(marks) > nsteps, "not enough steps in the trace"
lo, hi = marks[-nsteps - 1], marks[-1]
sel = rows[lo:hi]
span = sel[-1][0] - sel[0][0]
print(f"{nsteps} steps, {len(sel) / nsteps:.0f} kernels per step, {span / nsteps / 1e6:.3f} ms per step (xent to xent)")


def family(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)", k)
    return m.group(1) if m else k[:40]


fam = defaultdict(lambda: [0, 0])
for s, e, k, q in sel:
    f_ = family(k)
    fam[f_][0] += 1
    fam[f_][1] += e - s
print("-- kernel families, per step (launches, ms)")
for f_, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {f_:40s} {n / nsteps:7.1f}  {t / nsteps / 1e6:8.3f}")
print(f"  {'(sum)':40s} {len(sel) / nsteps:7.1f}  {sum(t for _, t in fam.values()) / nsteps / 1e6:8.3f}")
queues = defaultdict(list)
for r in sel:
    queues[r[3]].append(r)
main_q = max(queues, key=lambda q: len(queues[q]))
print("-- queues:", {q: (len(v) // nsteps, round(sum(e - s for s, e, _, _ in v) / nsteps / 1e6, 3)) for q, v in queues.items()},
      "main =", main_q)
mq = queues[main_q]
hist = defaultdict(lambda: [0, 0])
idle = 0
for a, b in zip(mq, mq[1:]):
    g = b[0] - a[1]
    if g <= 0:
        continue
    idle += g
    c = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-20us" if g < 20000 else ">20us"
    hist[c][0] += 1
    hist[c][1] += g
print(f"-- main queue: busy {sum(e - s for s, e, _, _ in mq) / nsteps / 1e6:.3f} ms, idle between its kernels {idle / nsteps / 1e6:.3f} ms per step")
for c in ("<1us", "1-2us", "2-4us", "4-8us", "8-20us", ">20us"):
    print(f"   gaps {c:7s}: {hist[c][0] / nsteps:7.1f} per step, {hist[c][1] / nsteps / 1e6:.3f} ms")
# the largest gaps: what ran before / after
big = sorted(((b[0] - a[1], a[2], b[2]) for a, b in zip(mq, mq[1:])), reverse=True)[: 12]
for g, ka, kb in big:
    print(f"   gap {g / 1e3:7.1f} us  after {family(ka)[:36]:36s} before {family(kb)[:36]}")
