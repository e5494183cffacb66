"""Debug helper (GPU box): per-parameter momentum buffer (= first-step gradient + wd*w) of the
fused TrainStep vs the CPU oracle, and the same after more steps."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from oracle import filler
from oracle import torch_ref as R
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep

name = sys.argv[1] if len(sys.argv) > 1 else "cspdarknet53"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
graphs = len(sys.argv) > 3 and sys.argv[3] == "graph"
ncls, B, S, lr, wd = 16, 8, 96, 2e-4, 1e-3
x, y = filler.images(B, S), filler.labels(B, ncls)

sd = {}
for k, shape in R.classifier_spec(name, ncls).items():
    dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
    sd[k] = filler.fill_tensor("tr." + k, torch.zeros(shape, dtype=dt))
params = {k: v for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
for v in params.values():
    v.requires_grad_(True)
mom = {}
for _ in range(steps):
    for v in params.values():
        v.grad = None
    loss, _ = R.classifier_loss(name, sd, x, y, 0.1, training=True)
    loss.backward()
    R.sgd_step(params, {k: v.grad for k, v in params.items()}, mom, lr, 0.9,
               lambda k: R.weight_decay_group(k, wd, 0.0, 0.0))
    print("oracle loss", loss.item())

ts = TrainStep(getattr(backbones, name)(), ncls, B, S, torch.float32, lr=lr, momentum=0.9, weight_decay=wd,
               label_smoothing=0.1, device="cuda", use_graphs=graphs)
filler.fill_module(ts.model, "tr.")
ts.weights_changed()
for _ in range(steps):
    ts.step(x.cuda(), y.cuda())
    print("hip loss", ts.loss())
torch.cuda.synchronize()
st = ts.store
for (k, p), off in zip(((k, p) for k, p in ts.model.named_parameters()), None or [None] * 10**6):
    base, o, n = st.where(p)
    m = ts.mflat[o : o + n].cpu()
    if p.dim() == 4:
        oo, ii, kh, kw = p.shape
        m = m.view(oo, kh, kw, ii).permute(0, 3, 1, 2)
    else:
        m = m.view(p.shape)
    ref = mom[k]
    err = ((m - ref).norm() / ref.norm().clamp_min(1e-12)).item()
    flag = " <<<" if err > 2e-2 else ""
    print(f"{err:10.3e}  {ref.norm().item():10.3e}  {k}{flag}")
