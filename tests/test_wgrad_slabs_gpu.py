"""vt_conv_wgrad_slabs: the filter gradient (autograd backward of nn.Conv2d w.r.t. its weight, components.py:26-35) in
two stages -- partial tiles stored per pixel split, then an ordered reducer -- against the atomic flush of
vt_conv_wgrad on the same operands, and against itself (bit-identical from call to call, which the atomic flush is
not).  Measured on the headline step: 23.9 ms with a 48 MiB scratch against 23.2 ms with atomics (66 extra launches),
so the engine keeps the atomics by default (VT_WGRAD_SLABS_MB=0); VT_DETERMINISTIC=1 takes this path."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import conv_desc, stream, vp

pytestmark = pytest.mark.gpu

# B, Cin, Cout, k, s, H   (general kernel 1x1 / stride 2 / small map; all-taps kernel 3x3)
CASES = [(32, 128, 128, 1, 1, 28), (16, 64, 128, 3, 2, 56), (64, 256, 256, 3, 1, 14), (32, 128, 128, 3, 1, 28),
         (16, 64, 64, 3, 1, 56)]


@pytest.mark.parametrize("B,Cin,Cout,k,s,H", CASES)
def test_two_stage_filter_gradient_equals_the_atomic_one_and_repeats_bitwise(B, Cin, Cout, k, s, H):
    torch.manual_seed(Cin + H)
    pad = -((s - k) // 2)
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    dz = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16)
    d = conv_desc(N.VT_BF16, x, Cin, Cout, k, s, pad, Cout)
    K = k * k * Cin
    lib = N.lib()
    ref = torch.full((Cout, K), 0.5, device="cuda")
    N.check(lib.vt_conv_wgrad(C.byref(d), vp(x), vp(dz), vp(ref), K, stream()))
    scratch = torch.empty(48 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(3):
        scratch.fill_(0xFF)  # NaN patterns: every slab element that is read must have been written
        dw = torch.full((Cout, K), 0.5, device="cuda")
        N.check(lib.vt_conv_wgrad_slabs(C.byref(d), vp(x), vp(dz), vp(dw), K, vp(scratch), scratch.numel(), stream()))
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.isfinite(outs[0]).all()
    assert ((outs[0] - ref).norm() / (ref - 0.5).norm()).item() < 2e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
