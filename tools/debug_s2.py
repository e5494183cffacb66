"""where do the stride-2 span kernel's outputs differ from the float64 conv? (debug)"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import torch
import torch.nn.functional as F
from vision_toolbox import _native as N
from gpu_util import TD, conv_desc, krsc, nhwc, rounded, stream, to_nchw, vp

B, Cin, Cout, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 32, 64, 56, 56))]
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W)
w = torch.randn(Cout, Cin, 3, 3) * 0.1
dt = N.VT_BF16
ref = F.conv2d(rounded(x, dt).double(), rounded(w, dt).double(), None, 2, 1)
xd, wd = nhwc(x, dt), krsc(w, dt)
y = torch.full((B, H // 2, W // 2, Cout), float("nan"), device="cuda", dtype=TD[dt])
d = conv_desc(dt, xd, Cin, Cout, 3, 2, 1, Cout)
N.set_knob("VT_SPAN_S2", 2)
N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), None, None, None, None, stream()))
torch.cuda.synchronize()
print(N.last_kernel_name())
got = to_nchw(y).double()
err = (got - ref).abs().amax(1)  # [B, Ho, Wo]
bad = (err > 0.05).nonzero()
print("bad positions", len(bad), "of", err.numel())
Wo = W // 2
for b, i, j in bad[:40].tolist():
    m = (b * (H // 2) + i) * Wo + j
    print(f"b{b} i{i} j{j} m{m} tile_row{m % 128} err{err[b, i, j]:.3f}")
# which single tap is missing / wrong?  compare with the conv minus one tap
for b, i, j in bad[:6].tolist():
    diffs = []
    for r in range(3):
        for s in range(3):
            w1 = rounded(w, dt).double().clone()
            w1[:, :, r, s] = 0
            alt = F.conv2d(rounded(x, dt).double()[b : b + 1], w1, None, 2, 1)[0, :, i, j]
            diffs.append(((alt - got[b, :, i, j]).abs().max().item(), r, s))
    print((b, i, j), "closest when dropping tap", min(diffs))
