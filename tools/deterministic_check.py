"""VT_DETERMINISTIC=1: two fresh train runs (same seed, same data, two SGD steps with momentum) must end with
BIT-IDENTICAL parameters, gradients, momentum and BatchNorm state.  What makes that possible: the BatchNorm statistics
and backward sums are always fixed-point integer atomics (vt_common.h), and in this mode the filter gradients are
two-stage (vt_conv_wgrad_slabs: stored partial tiles + an ordered reducer), the bias column sums go through a
fixed-point shadow (vt_colsum_fixed, vt_fixed_to_f32) and the one-pass stem kernel accumulates in fixed point.  (The scalar
loss is still a float atomic over the batch rows: it is reported, nothing is computed from it.)

    VT_DETERMINISTIC=1 python tools/deterministic_check.py [model] [f32|bf16]        (GPU box)"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


def run(model, dt, B, S):
    torch.manual_seed(0)
    ts = TrainStep(getattr(backbones, model)(), 16, B, S, dt, lr=0.02, momentum=0.9, use_graphs=False)
    assert ts.prog.builder.deterministic == (os.environ.get("VT_DETERMINISTIC", "0") != "0")
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        x = torch.randn(B, 3, S, S, generator=g).cuda()
        y = torch.randint(0, 16, (B,), generator=g).cuda()
        ts.step(x, y)
    torch.cuda.synchronize()
    return [t.clone() for t in (ts.store.pflat, ts.gflat, ts.mflat, ts.store.sflat)], ts.prog.kind_histogram


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "darknet_yolov5n"
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
    B, S = (16, 64) if model != "vovnet19_slim_ese" else (8, 64)
    a, hist = run(model, dt, B, S)
    ok = True
    for r in range(3):
        b, _ = run(model, dt, B, S)
        same = [bool(torch.equal(u, v)) for u, v in zip(a, b)]
        rel = [float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30)) for u, v in zip(a, b)]
        print(f"run {r + 1}: params/grads/momentum/bn-state identical {same} (rel diff {['%.1e' % v for v in rel]})", flush=True)
        ok = ok and all(same)
    print("ops:", {k: v for k, v in hist.items() if k in ("conv_wgrad", "fixed_to_f32", "colsum", "stem_bwd_reduce")}, "slab MiB", os.environ.get("VT_WGRAD_SLABS_MB", "48 (default in this mode)"))
    det = os.environ.get("VT_DETERMINISTIC", "0") != "0"
    if det:
        assert ok, "deterministic mode produced different bits"
        print("DETERMINISTIC_OK")
    else:
        print("(default mode: equality not required)")


if __name__ == "__main__":
    main()
