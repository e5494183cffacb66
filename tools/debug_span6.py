"""dev: where span6 differs from the span kernel on one shape.  python tools/debug_span6.py B Cin Cout H W [slices]"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import torch

from vision_toolbox import _native as N
from test_span6_gpu import _desc, _run

B, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
slices = len(sys.argv) > 6 and sys.argv[6] == "1"
xoff = int(sys.argv[7]) if len(sys.argv) > 7 else 16
flags = {"stats": N.VT_CONV_STATS, "plain": 0}[sys.argv[8]] if len(sys.argv) > 8 else 0
only = sys.argv[9] if len(sys.argv) > 9 else ""
torch.manual_seed(1)
ldx, ldy = (Cin + 32, Cout + 64) if slices else (Cin, Cout)
xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
x = xb[..., xoff:xoff + Cin] if slices else xb
w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
d = _desc(B, Cin, Cout, H, W, ldx, ldy, 0, flags, False)
st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
ys = []
for env in (("0", "2") if not only else (only,)):
    yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = yb[..., 32:32 + Cout] if slices else yb
    name = _run(env, d, x, w, y, None, None, None, st)
    ys.append(y.float().clone())
    print(env, name)
if len(ys) < 2:
    print('ran', only); sys.exit(0)
a, b = ys
bad = (a != b) | (torch.isnan(a) != torch.isnan(b))
print("mismatching elements", int(bad.sum()), "of", bad.numel(), "nan in span6 out", int(torch.isnan(b).sum()))
if bad.any():
    pix = bad.reshape(-1, Cout).any(1).nonzero().flatten()
    ch = bad.reshape(-1, Cout).any(0).nonzero().flatten()
    print("bad pixels", pix.numel(), "first", pix[:12].tolist(), "last", pix[-5:].tolist())
    print("bad channels", ch.numel(), ch[:16].tolist())
    p0 = int(pix[0])
    print("pixel", p0, "(b,i,j) =", p0 // (H * W), (p0 // W) % H, p0 % W, "span", a.reshape(-1, Cout)[p0, :6].tolist(), "span6", b.reshape(-1, Cout)[p0, :6].tolist())
    # per-image-column / row histogram
    jj = (pix % W).bincount(minlength=W)
    ii = ((pix // W) % H).bincount(minlength=H)
    print("bad by column j:", jj.tolist())
    print("bad by row i:", ii.tolist())
