"""Generate tests/golden/necks.npz by running the UNMODIFIED reference necks (FPN / PAN,
vision_toolbox/necks.py) on CPU.  Same shim as tools/gen_golden.py; run in the build container.

    python tools/gen_golden_necks.py
"""
from __future__ import annotations

import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import gen_golden  # noqa: E402  (the import shim for the reference's vision_toolbox.*)
from oracle import filler  # noqa: E402

necks = gen_golden.ref_import("vision_toolbox.necks")
GOLDEN = ROOT / "tests" / "golden"

CASES = {  # name -> (kind, in_channels, out_channels, top_down, sizes(bottom first), batch[, fuse_fn])
    "fpn_td": ("fpn", (16, 32, 64), 32, True, (16, 8, 4), 2),
    "fpn_bu": ("fpn", (16, 32, 64), 32, False, (16, 8, 4), 2),
    "pan": ("pan", (16, 24, 40), 16, True, (16, 8, 4), 2),
    # round 6: fuse_fn="concat" (necks.py:14-15, 66: the output conv takes 2 x out_channels)
    "fpn_td_cat": ("fpn", (16, 32, 64), 32, True, (16, 8, 4), 2, "concat"),
    "fpn_bu_cat": ("fpn", (16, 32, 64), 32, False, (16, 8, 4), 2, "concat"),
    "pan_cat": ("pan", (16, 24, 40), 16, True, (16, 8, 4), 2, "concat"),
    # interpolation_mode="bilinear" (nn.Upsample, necks.py:65), both directions
    "fpn_td_bil": ("fpn", (16, 32, 64), 32, True, (16, 8, 4), 2, "sum", "bilinear"),
    "fpn_bu_bil": ("fpn", (16, 32, 64), 32, False, (16, 8, 4), 2, "sum", "bilinear"),
    "pan_cat_bil": ("pan", (16, 24, 40), 16, True, (16, 8, 4), 2, "concat", "bilinear"),
}


def np_(t):
    return t.detach().cpu().numpy().copy()


def main():
    out = {}
    for name, case in CASES.items():
        kind, ins, outc, td, sizes, B = case[:6]
        fuse = case[6] if len(case) > 6 else "sum"
        interp = case[7] if len(case) > 7 else "nearest"
        torch.manual_seed(0)
        m = (necks.FPN(list(ins), outc, fuse_fn=fuse, interpolation_mode=interp, top_down=td) if kind == "fpn"
             else necks.PAN(list(ins), outc, fuse_fn=fuse, interpolation_mode=interp))
        filler.fill_module(m, f"{name}.")
        out[f"{name}/keys"] = np.array(list(m.state_dict().keys()))
        out[f"{name}/shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
        for mode in ("train", "eval"):
            filler.fill_module(m, f"{name}.")
            m.train(mode == "train")
            xs = [filler.tensor(f"{name}.x{i}", (B, c, s, s)).requires_grad_(True) for i, (c, s) in enumerate(zip(ins, sizes))]
            ys = m([x for x in xs])
            loss = sum((y * filler.tensor(f"{name}.r{i}", tuple(y.shape))).sum() for i, y in enumerate(ys))
            m.zero_grad()
            loss.backward()
            for i, y in enumerate(ys):
                out[f"{name}/{mode}/y{i}"] = np_(y)
            for i, x in enumerate(xs):
                out[f"{name}/{mode}/dx{i}"] = np_(x.grad)
            for k, p in m.named_parameters():
                out[f"{name}/{mode}/grad/{k}"] = np_(p.grad)
            if mode == "train":
                for k, v in m.state_dict().items():
                    if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                        out[f"{name}/train/state/{k}"] = np_(v)
    np.savez_compressed(GOLDEN / "necks.npz", **out)
    print(f"wrote {GOLDEN / 'necks.npz'}: {len(out)} arrays")


if __name__ == "__main__":
    main()
