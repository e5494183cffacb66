"""The data-parallel schedule of TrainStep captured as hipGraphs against the same schedule run eagerly.

The trainer cuts the backward launch list behind every op that completes a gradient bucket.  The engine releases the held
filter gradients of a stage as ONE fork followed by up to eight side-stream ops, so a cut can land inside such a run: the
next segment then BEGINS with side-stream ops whose fork sits in the previous segment.  Eagerly the in-order side stream
carries the dependency; a per-segment capture must fork by itself (vt_runtime.hip, `capture_fork`), else those launches
run once at capture time and their filter gradients are zero on every replay.

    python tools/dp_graph_check.py [bucket_mb]          (GPU box; a one-rank gloo group + VT_DP_WORLD1=1)

Checks, at bucket sizes that put cuts inside the held runs: (1) the plan really has backward segments whose first
side-stream op has no FORK in front of it inside the segment; (2) every parameter gradient of two consecutive bf16 steps
(lr = 0) equals the eager schedule's up to the order of the f32 atomic sums."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch
import torch.distributed as dist

from oracle import filler
from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


def orphan_segments(ts) -> int:
    """backward segments whose first side-stream op is not preceded by a FORK / FORK_WAIT inside the segment"""
    n, lo = 0, 0
    for hi in ts.bwd_cuts:
        forked = False
        for i in range(lo, hi):
            kind = ts.prog.bwd_ops[i].kind
            base = kind & 0xFFFF
            if base in (N.OP_FORK, N.OP_FORK_WAIT):
                forked = True
            elif kind & N.OP_SIDE_STREAM:
                n += 0 if forked else 1
                break
        lo = hi
    return n


def grads(graphs: bool, bucket_mb: float, x, y):
    ts = TrainStep(backbones.cspdarknet53(), 16, x.shape[0], x.shape[-1], torch.bfloat16, lr=0.0, momentum=0.0,
                   weight_decay=0.0, label_smoothing=0.1, device="cuda", use_graphs=graphs, bucket_mb=bucket_mb)
    assert ts.dp and ts.bucketer is not None and len(ts.bwd_cuts) > 2
    filler.fill_module(ts.model, "dpg.")
    ts.weights_changed()
    ts.broadcast_parameters(0)
    out = []
    for _ in range(2):  # the second replay too
        ts.step(x, y)
        torch.cuda.synchronize()
        out.append(ts.gflat.double().clone())
    return ts, out


def main():
    bucket_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    os.environ["VT_DP_WORLD1"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    x, y = filler.images(8, 96).cuda(), filler.labels(8, 16).cuda()
    ts_e, g_e = grads(False, bucket_mb, x, y)
    orphans = orphan_segments(ts_e)
    print(f"bucket_mb {bucket_mb}: {len(ts_e.bwd_cuts)} backward segments, {orphans} begin with side-stream ops and no fork")
    assert orphans > 0, "this bucket size does not cut inside a held run: the check would prove nothing"
    ts_g, g_g = grads(True, bucket_mb, x, y)
    assert ts_g._graphs is not None and len(ts_g._graphs["bwd"]) == len(ts_g.bwd_cuts)
    worst = 0.0
    for step in range(2):
        for p, off in zip(ts_e.store.params, ts_e.store.offsets):
            a, b = g_e[step][off : off + p.numel()], g_g[step][off : off + p.numel()]
            if float(a.norm()) == 0.0:
                continue
            e = float((a - b).norm() / a.norm())
            worst = max(worst, e)
            assert e < 1e-4, f"step {step}: gradient at offset {off} ({tuple(p.shape)}) differs by {e:.3e} (graph norm {float(b.norm()):.3e})"
    print(f"captured vs eager data-parallel schedule: worst per-parameter gradient difference {worst:.2e}")
    dist.destroy_process_group()
    print("DP_GRAPH_OK")


if __name__ == "__main__":
    main()
