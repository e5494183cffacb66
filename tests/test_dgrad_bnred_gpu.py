"""vt_conv_dgrad_bnred (round 6): a 3x3 stride-1 data gradient that also forms the BatchNorm-backward sums of the unit whose
output it differentiates -- the autograd backward of nn.Conv2d with respect to its input (reference components.py:26-35) plus
the two per-channel reductions of the autograd backward of nn.BatchNorm2d / nn.ReLU (components.py:36-44) of the unit in
front of it (DarknetBlock.conv1 <- conv2, darknet.py:23-28).

Checked on identical operands:
  * d(y) is BIT-IDENTICAL to what vt_conv_igemm stores for the same descriptor (the fused epilogue only adds to what the
    plain one does);
  * the sums equal (a) those of the separate pass vt_bn_act_bwd_reduce over that d(y) and z -- same per-element f32
    arithmetic, another summation order: 1e-5 of the column's scale -- and (b) a float64 evaluation of
    sum g, invstd * sum g (z - mean), g = d(y) * [z * scale + shift > 0], on the stored bf16 values;
  * shapes: the three dominant CSPDarknet-53 geometries (padded rows, pixel rows, K-split), a 160-column layer (the
    128 + 32 split cannot carry the sums: falls back to two launches), a shape span6 does not take at all, channel-slice
    operands, with and without the ReLU mask."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# B, C (in = out), H, W, expect the fused kernel
SHAPES = [
    (96, 128, 28, 28, True),    # the dominant layer's geometry (padded positions)
    (256, 256, 14, 14, True),   # pixel rows (masked)
    (128, 256, 14, 14, True),   # K-split at the data-parallel per-GPU batch
    (256, 512, 7, 7, True),     # K-split
    (40, 64, 45, 37, True),     # odd sizes, one filter tile half full
    (24, 160, 28, 28, False),   # 128 + 32 columns: no fused form, two launches
    (64, 32, 28, 28, False),    # a single channel chunk: not a span6 launch
]


def _desc(B, Cin, Cout, H, W, ldx, ldy):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, ldx
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, W, 1, 1, -1, -1
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, ldy, H, W, 1, 1
    d.ldw, d.ldr, d.flags, d.ntaps = 9 * Cin, 0, 0, 9
    for i in range(9):
        t = 8 - i  # the tap order of a stride-1 data gradient
        d.dh[i], d.dw[i] = t // 3, t % 3
    return d


@pytest.mark.parametrize("relu", [1, 0], ids=["relu", "no_act"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[:4])))
def test_fused_sums_match_the_separate_reduction_and_float64(shape, relu):
    B, Cc, H, W, fused = shape
    torch.manual_seed(Cc + H)
    slices = B % 16 == 0 and Cc >= 64  # channel-slice operands on some shapes
    ldx, ldy, ldz = (Cc + 32, Cc + 64, Cc + 16) if slices else (Cc, Cc, Cc)
    dzb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    dz = dzb[..., 16:16 + Cc] if slices else dzb
    w = (torch.randn(Cc, 9, Cc, device="cuda") * (2.0 / (9 * Cc)) ** 0.5).to(torch.bfloat16)
    zb = (torch.randn(B, H, W, ldz, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    z = zb[..., 8:8 + Cc] if slices else zb
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.3
    mean = z.float().mean((0, 1, 2))
    invstd = 1.0 / torch.sqrt(z.float().var((0, 1, 2), unbiased=False) + 1e-5)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    d = _desc(B, Cc, Cc, H, W, ldx, ldy)
    lib = N.lib()
    N.set_knob("VT_SPAN6", 2)  # (test sizes are below the dispatcher's own thresholds)
    try:
        # reference: the two launches
        yb0 = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
        y0 = yb0[..., 32:32 + Cc] if slices else yb0
        N.check(lib.vt_conv_igemm(C.byref(d), vp(dz), vp(w), vp(y0), None, None, None, None, stream()))
        s0 = N.stats_buffer(Cc)
        N.check(lib.vt_bn_act_bwd_reduce(vp(y0), ldy, vp(z), ldz, vp(scale), vp(shift), vp(mean), vp(invstd), B * H * W, Cc, relu,
                                         N.VT_BF16, vp(s0), stream()))
        # the fused call
        yb1 = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
        y1 = yb1[..., 32:32 + Cc] if slices else yb1
        s1 = N.stats_buffer(Cc)
        before = N.launch_count()
        N.check(lib.vt_conv_dgrad_bnred(C.byref(d), vp(dz), vp(w), vp(y1), vp(z), ldz, vp(scale), vp(shift), vp(mean),
                                        vp(invstd), relu, vp(s1), stream()))
        torch.cuda.synchronize()
        launches = N.launch_count() - before
        name = N.last_kernel_name()
    finally:
        N.set_knob("VT_SPAN6", 1)
    assert (launches == 1) == fused, (launches, name)  # one launch where the fused kernel exists, the two it replaces elsewhere
    if fused:
        assert "span6" in name, name
    assert torch.equal(torch.isnan(yb0.float()), torch.isnan(yb1.float()))  # nothing outside the slice was written
    assert torch.equal(torch.nan_to_num(yb0.float()), torch.nan_to_num(yb1.float()))  # d(y): bit-identical
    a, b = N.stats_decode(s0), N.stats_decode(s1)
    dy64, z64 = y1.double(), z.double()
    mask = (torch.addcmul(shift, z.float(), scale) > 0).double() if relu else torch.ones_like(z64)
    g = dy64 * mask
    ref = torch.stack([g.sum((0, 1, 2)), (g * (z64 - mean.double())).sum((0, 1, 2)) * invstd.double()])
    col = torch.stack([g.abs().sum((0, 1, 2)), (g * (z64 - mean.double())).abs().sum((0, 1, 2)) * invstd.double()])
    # against the separate pass: the same f32 terms in another order; against float64: f32 partial sums of <= 112 rows
    assert ((b - a).abs() / col).max().item() < 1e-5
    assert ((b - ref).abs() / col).max().item() < 2e-5
    assert ((a - ref).abs() / col).max().item() < 2e-5
