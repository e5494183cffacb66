"""run-to-run spread of the first step's gradients, per parameter (race vs. sensitivity)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep
model = os.environ.get("MODEL", "darknet_yolov5n")
B, S = int(os.environ.get("B", "8")), int(os.environ.get("S", "64"))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, S, S, generator=g).cuda()
y = torch.randint(0, 16, (B,), generator=g).cuda()
grads, names = [], None
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    torch.manual_seed(0)
    ts = TrainStep(getattr(backbones, model)(), 16, B, S, torch.float32 if os.environ.get("F32") else torch.bfloat16, lr=0.0, use_graphs=False)
    ts.step(x, y)
    torch.cuda.synchronize()
    grads.append(ts.gflat.double().clone())
    if names is None:
        st = ts.store
        named = {id(p): n for n, p in ts.model.named_parameters()}
        names = [(named[id(p)], off, p.numel()) for p, off in zip(st.params, st.offsets)]
    print(i, ts.loss(), flush=True)
    del ts
ref = grads[0]
for k, gk in enumerate(grads[1:], 1):
    worst = []
    for n, off, cnt in names:
        a, b = ref[off:off + cnt], gk[off:off + cnt]
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        worst.append((rel, n, float(a.norm())))
    worst.sort(reverse=True)
    print(f"run {k} vs 0: total rel {float((ref - gk).norm() / ref.norm()):.2e}; worst:", [(f"{r:.1e}", n, f"{nm:.1e}") for r, n, nm in worst[:5]])
