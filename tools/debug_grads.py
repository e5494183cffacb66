"""Debug helper (GPU box): per-parameter gradient error of the HIP path vs the CPU oracle.

    python tools/debug_grads.py darknet_yolov5n eval|train [f32|bf16] [batch] [size]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch
import torch.nn.functional as F

from oracle import filler
from oracle import torch_ref as R
from vision_toolbox import backbones

name = sys.argv[1] if len(sys.argv) > 1 else "darknet_yolov5n"
mode = sys.argv[2] if len(sys.argv) > 2 else "eval"
dt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
S = int(sys.argv[5]) if len(sys.argv) > 5 else 64
training = mode == "train"

bb = getattr(backbones, name)()
model = torch.nn.Sequential(bb, torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                            torch.nn.Linear(bb.get_last_out_channels(), 16))
filler.fill_module(model, name + ".")
sd = {k: v.clone() for k, v in model.state_dict().items()}
for k, v in sd.items():
    if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
        v.requires_grad_(True)
x, y = filler.images(B, S), filler.labels(B, 16)
loss, logits = R.classifier_loss(name, sd, x, y, 0.1, training=training)
loss.backward()

bb.compute_dtype = dt
model = model.cuda().train(training)
out = model[0](x.cuda())
lg = model[3](model[2](model[1](out.float())))
l2 = F.cross_entropy(lg, y.cuda(), label_smoothing=0.1)
l2.backward()
print("loss", loss.item(), l2.item())
for k, p in model.named_parameters():
    ref = sd[k].grad
    err = ((p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-12)).item()
    flag = " <<<" if err > 1e-3 else ""
    print(f"{err:10.3e}  {k}{flag}")
