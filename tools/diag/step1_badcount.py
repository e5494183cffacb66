"""how often does a fresh f32 train step produce gradients off the majority class?"""
import os, sys
from pathlib import Path
ROOT = Path(os.environ.get("VT_TREE", Path(__file__).resolve().parents[2]))
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep
dt = torch.float32 if os.environ.get("F32", "1") != "0" else torch.bfloat16
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 3, 64, 64, generator=g).cuda(); y = torch.randint(0, 16, (8,), generator=g).cuda()
gs = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    torch.manual_seed(0)
    ts = TrainStep(getattr(backbones, os.environ.get("MODEL", "darknet_yolov5n"))(), 16, 8, 64, dt, lr=0.0, use_graphs=False)
    ts.step(x, y); torch.cuda.synchronize()
    gs.append(ts.gflat.double().clone()); del ts
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
d = [rel(a, gs[0]) for a in gs[1:]]
bad = sum(1 for v in d if v > 1e-4)
print(os.environ.get("TAG", ""), "bad", bad, "of", len(d), "max", f"{max(d):.1e}", flush=True)
