import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import numpy as np, torch
from oracle import filler
import test_modules_gpu as T
name = "vovnet39"
ref = None
for i in range(10):
    model = T._classifier(name, torch.bfloat16)
    x = filler.images(4, 64).cuda()
    model.train()
    with torch.no_grad() if os.environ.get("NOGRAD") else torch.enable_grad():
        maps = model[0].get_feature_maps(x)
    maps = [m.float().clone() for m in maps]
    torch.cuda.synchronize()
    if ref is None:
        ref = maps
    print(i, [f"{float((a - b).norm() / b.norm()):.2e}" for a, b in zip(maps, ref)], [f"{float(m.norm()):.4f}" for m in maps], flush=True)
