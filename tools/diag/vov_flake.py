import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import numpy as np, torch, torch.nn.functional as F
from oracle import filler
import test_modules_gpu as T
gm = np.load(ROOT / "tests" / "golden" / "models.npz")
name = "vovnet39"
res = {}
for i in range(8):
    model = T._classifier(name, torch.bfloat16)
    x, y = filler.images(4, 64).cuda(), filler.labels(4, 16).cuda()
    model.train()
    logits = model[3](model[2](model[1](model[0](x).float())))
    loss = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss.backward()
    keys = list(gm[f"{name}.train.grad_keys"]); norms = gm[f"{name}.train.grad_norms"]
    params = dict(model.named_parameters())
    got = np.array([params[k].grad.double().norm().item() for k in keys])
    big = norms > 1e-3 * norms.max()
    med = round(float(np.median(np.abs(got[big] / norms[big] - 1))), 4)
    print(i, med, float(loss), flush=True)
    res.setdefault(med, (got, [params[k].grad.double().clone() for k in keys]))
if len(res) >= 2:
    (ma, (ga, fa)), (mb, (gb, fb)) = list(res.items())[:2]
    print("classes", ma, mb)
    rows = []
    for k, a, b, n in zip(keys, fa, fb, norms):
        rows.append((float((a - b).norm() / (b.norm() + 1e-30)), k, float(a.norm()), float(b.norm()), float(n)))
    for r in rows:
        if r[0] > 1e-3:
            print(f"{r[0]:.3e} {r[1]:50s} |a| {r[2]:.4e} |b| {r[3]:.4e} ref {r[4]:.4e}")
