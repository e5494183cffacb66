"""Dev (GPU box): conditioning of the filter gradient of a train-mode bf16 ConvNormAct unit when its input has a mean
(post-ReLU activations) -- the single-unit test of tests/test_fullsize_gpu.py with x = relu(randn) instead of randn."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import conftest  # noqa: F401
import torch

import test_fullsize_gpu as T
from vision_toolbox.components import ConvNormAct

B, Cin, Cout, k, s, H = 256, 128, 128, 1, 1, 56
for name, make in (("randn", lambda g: torch.randn(B, Cin, H, H, generator=g)),
                   ("relu(randn)", lambda g: torch.relu(torch.randn(B, Cin, H, H, generator=g)))):
    for gmean in (0.25, 0.0):
        torch.manual_seed(1)
        m = ConvNormAct(Cin, Cout, k, s)
        with torch.no_grad():
            m.conv.weight.copy_(m.conv.weight.to(torch.bfloat16).float())
            m.norm.weight.uniform_(0.5, 1.5)
            m.norm.bias.uniform_(-0.3, 0.3)
        w0, g0, b0 = m.conv.weight.detach().clone(), m.norm.weight.detach().clone(), m.norm.bias.detach().clone()
        gen = torch.Generator().manual_seed(2)
        x = make(gen).to(torch.bfloat16).float()
        gy = (torch.randn(B, Cout, H, H, generator=gen) + gmean).to(torch.bfloat16).float()
        refs = {bs: T._unit_reference(x, w0, g0, b0, gy, s, 0, bs) for bs in (True, False)}
        m = m.cuda().train()
        m.compute_dtype = torch.bfloat16
        xg = x.cuda().requires_grad_(True)
        y = m(xg)
        y.backward(gy.cuda().to(y.dtype))
        torch.cuda.synchronize()
        rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
        ry, rdx, rdw, rdg, rdb, _ = refs[True]
        fy, fdx, fdw, fdg, fdb, _ = refs[False]
        print(f"x={name:12s} gy mean {gmean}: ours vs bf16-storage ref: dw {rel(m.conv.weight.grad, rdw):.2e} dx {rel(xg.grad, rdx):.2e} "
              f"dgamma {rel(m.norm.weight.grad, rdg):.2e} | bf16-storage ref vs pure f64: dw {rel(rdw, fdw):.2e} dx {rel(rdx, fdx):.2e} "
              f"dgamma {rel(rdg, fdg):.2e}", flush=True)
