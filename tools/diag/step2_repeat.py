"""two SGD steps, fresh instances: where do the parameters of run k leave those of run 0?"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep
dt = torch.float32 if os.environ.get("F32", "1") != "0" else torch.bfloat16
g = torch.Generator().manual_seed(1)
xs = [torch.randn(8, 3, 64, 64, generator=g).cuda() for _ in range(2)]
ys = [torch.randint(0, 16, (8,), generator=g).cuda() for _ in range(2)]
snaps = []
names = None
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.manual_seed(0)
    ts = TrainStep(backbones.darknet_yolov5n(), 16, 8, 64, dt, lr=0.01, use_graphs=False)
    out = []
    for k in range(2):
        ts.step(xs[k], ys[k])
        torch.cuda.synchronize()
        out.append((ts.loss(), ts.store.pflat.double().clone(), ts.gflat.double().clone(), ts.store.sflat.double().clone(), ts.mflat.double().clone()))
    snaps.append(out)
    if names is None:
        st = ts.store
        named = {id(p): n for n, p in ts.model.named_parameters()}
        names = [(named[id(p)], off, p.numel()) for p, off in zip(st.params, st.offsets)]
    del ts
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
for r in range(1, len(snaps)):
    line = []
    for k in range(2):
        l, p, gg, s, m = snaps[r][k]; l0, p0, g0, s0, m0 = snaps[0][k]
        line.append(f"step{k+1}: dloss {abs(l-l0):.1e} params {rel(p,p0):.1e} grads {rel(gg,g0):.1e} bnstate {rel(s,s0):.1e} mom {rel(m,m0):.1e}")
    print(r, " | ".join(line), flush=True)
    if rel(snaps[r][0][4], snaps[0][0][4]) > 1e-4:
        mm, m0 = snaps[r][0][4], snaps[0][0][4]
        w = sorted(((rel(mm[o:o+c], m0[o:o+c]), n, float(m0[o:o+c].norm())) for n, o, c in names), reverse=True)[:5]
        print("   worst step-1 momentum:", [(f"{a:.1e}", n, f"{nm:.2e}") for a, n, nm in w])
        gg, g0 = snaps[r][0][2], snaps[0][0][2]
        w = sorted(((rel(gg[o:o+c], g0[o:o+c]), n, float(g0[o:o+c].norm())) for n, o, c in names), reverse=True)[:3]
        print("   worst step-1 grads (read after the step):", [(f"{a:.1e}", n, f"{nm:.2e}") for a, n, nm in w])
    if rel(snaps[r][1][2], snaps[0][1][2]) > 1e-4:
        gg, g0 = snaps[r][1][2], snaps[0][1][2]
        w = sorted(((rel(gg[o:o+c], g0[o:o+c]), n) for n, o, c in names), reverse=True)[:6]
        print("   worst step-2 grads:", [(f"{a:.1e}", n) for a, n in w])
        print("   in parameter order:", [(f"{rel(gg[o:o+c], g0[o:o+c]):.0e}", n.replace("0.stages.", "s").replace(".conv.weight", ".w").replace(".norm.weight", ".g").replace(".norm.bias", ".b")) for n, o, c in names])
        gg, g0 = snaps[r][0][1], snaps[0][0][1]
        w = sorted(((rel(gg[o:o+c], g0[o:o+c]), n) for n, o, c in names), reverse=True)[:4]
        print("   worst step-1 params:", [(f"{a:.1e}", n) for a, n in w])
