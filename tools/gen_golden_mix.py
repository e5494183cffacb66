"""Generate tests/golden/mix.npz by running the UNMODIFIED reference RandomCutMixMixUp
(/root/reference/extras.py:96-109) on CPU under fixed torch seeds.

extras.py imports `torchvision.transforms.functional` (absent here) for ONE helper,
`get_image_size`, whose documented behaviour is `[width, height]` of the tensor; the shim below
supplies exactly that and nothing else.  Run in the build container:

    python tools/gen_golden_mix.py
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import filler  # noqa: E402

tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
tr = types.ModuleType("torchvision.transforms")
tf = types.ModuleType("torchvision.transforms.functional")
tf.get_image_size = lambda img: [int(img.shape[-1]), int(img.shape[-2])]
tr.functional = tf
tv.transforms = tr
sys.modules["torchvision.transforms"] = tr
sys.modules["torchvision.transforms.functional"] = tf
spec = importlib.util.spec_from_file_location("ref_extras", "/root/reference/extras.py")
extras = importlib.util.module_from_spec(spec)
spec.loader.exec_module(extras)

B, NCLS, H, W = 4, 10, 12, 16


def main():
    out = {"meta": np.array([B, NCLS, H, W])}
    m = extras.RandomCutMixMixUp(NCLS, 1.0, 0.2)
    x, y = filler.tensor("mix.x", (B, 3, H, W)), filler.labels(B, NCLS, seed=77)
    for seed in range(12):
        torch.manual_seed(seed)
        xb, tb = m(x, y)
        out[f"s{seed}/images"] = xb.numpy().copy()
        out[f"s{seed}/target"] = tb.numpy().copy()
    np.savez_compressed(ROOT / "tests" / "golden" / "mix.npz", **out)
    print("wrote tests/golden/mix.npz")


if __name__ == "__main__":
    main()
