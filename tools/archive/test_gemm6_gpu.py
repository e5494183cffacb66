# ARCHIVED with tools/archive/vt_gemm6.hip (NOTEBOOK R5.13): passed 50 / 50 on the GPU when the kernel was in the build.
"""vt_gemm6.hip: the persistent two-group GEMM kernel of the big-K 1x1 stride-1 layers (the concat convolution of VoVNet's
OSA blocks, reference vovnet.py:50-63, its data gradient, the wide 1x1 units of Darknet-YOLOv5x, darknet.py:127-141)
against (a) the float64 product of the SAME bf16 operands -- one bf16 rounding of the output, two where a residual is
added -- and (b) the kernel vt_conv_igemm takes without it, on identical operands.

The model-level tests run toy sizes that never reach this kernel by default (it wants >= 256 input channels and 4M
outputs), so the dispatch is forced (VT_GEMM6=2) over a matrix of edge cases: pixel counts that are not multiples of 16
(rows past M come from the zero page and are never stored), every tile height (FM 4..7, by M and by the VT_GEMM6_FM
knob), filter-tile counts from 1 to 9 with partial last tiles (Cout = 72, 200, 1056), 2 .. 59 K-steps, fewer items than
workgroups and several items per workgroup, channel-slice operands (ldx > Cin, ldy > Cout, ldr > Cout), every epilogue."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# B, H, W, Cin, Cout
SHAPES = [
    (16, 56, 56, 768, 256),    # VoVNet-39 stage 2 concat at 1/16 of the batch
    (9, 28, 28, 1056, 512),    # stage 3: 33 K-steps, 4 filter tiles + ...
    (7, 14, 14, 1472, 768),    # stage 4 (M = 1372: not a multiple of 16; fewer items than workgroups)
    (32, 7, 7, 1888, 1024),    # stage 5: 59 K-steps
    (16, 56, 56, 256, 768),    # the stage-2 data gradient: 8 K-steps, six filter tiles
    (5, 33, 17, 64, 72),       # two K-steps, one partial filter tile
    (3, 41, 29, 96, 200),      # three K-steps, partial second tile
    (64, 20, 20, 640, 1056),   # nine filter tiles (do not divide 32), a 32-column tail
    (2, 160, 160, 160, 160),   # YOLOv5x C3 1x1 at 2 images: many items per workgroup
]
MODES = [("stats", N.VT_CONV_STATS), ("plain", 0), ("residual", N.VT_CONV_RESIDUAL),
         ("affine_relu", N.VT_CONV_AFFINE | N.VT_CONV_RELU),
         ("affine_relu_residual", N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL)]


def _desc(B, H, W, Cin, Cout, ldx, ldy, ldr, flags):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, ldx
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, W, 1, 1, 0, 0
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, ldy, H, W, 1, 1
    d.ldw, d.ldr, d.flags, d.ntaps = Cin, ldr, flags, 1
    d.dh[0] = d.dw[0] = 0
    return d


def _run(knob, d, x, w, y, scale, shift, res, stats, fm=0):
    N.set_knob("VT_GEMM6", knob)
    N.set_knob("VT_GEMM6_FM", fm)
    try:
        N.check(N.lib().vt_conv_igemm(C.byref(d), vp(x), vp(w), vp(y), vp(scale) if scale is not None else None,
                                      vp(shift) if shift is not None else None, vp(res) if res is not None else None,
                                      vp(stats) if stats is not None else None, stream()))
        torch.cuda.synchronize()
        return N.last_kernel_name()
    finally:
        N.set_knob("VT_GEMM6", 1)
        N.set_knob("VT_GEMM6_FM", 0)


def _operands(shape, flags, slices, seed):
    B, H, W, Cin, Cout = shape
    torch.manual_seed(seed)
    ldx, ldy, ldr = (Cin + 32, Cout + 64, Cout + 8) if slices else (Cin, Cout, Cout)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin] if slices else xb
    w = (torch.randn(Cout, Cin, device="cuda") * (2.0 / Cin) ** 0.5).to(torch.bfloat16)
    res = None
    if flags & N.VT_CONV_RESIDUAL:
        rb = torch.randn(B, H, W, ldr, device="cuda").to(torch.bfloat16)
        res = rb[..., 8:8 + Cout] if slices else rb
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, H, W, Cin, Cout, ldx, ldy, ldr if res is not None else 0, flags)
    return d, x, w, res, scale, shift, ldy


def _expected(x, w, res, scale, shift, flags):
    z = x.double().reshape(-1, x.shape[-1]) @ w.double().t()
    if flags & N.VT_CONV_AFFINE:
        z = z * scale.double() + shift.double()
    if flags & N.VT_CONV_RELU:
        z = z.clamp_min(0)
    z = z.reshape(*x.shape[:3], -1)
    if res is not None:  # the kernels round the epilogue's value to bf16 first, then add the residual in f32 and round again
        z = z.to(torch.bfloat16).double() + res.double()
    return z


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gemm6_against_float64_and_the_kernel_it_replaces(shape, mode):
    B, H, W, Cin, Cout = shape
    flags = mode[1]
    slices = (B + Cin // 32) % 2 == 0  # about half of the shapes: operands are channel slices of wider buffers
    d, x, w, res, scale, shift, ldy = _operands(shape, flags, slices, seed=sum(shape))
    outs = {}
    for knob in (0, 2):
        yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
        y = yb[..., 32:32 + Cout] if slices else yb
        st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
        name = _run(knob, d, x, w, y, scale, shift, res, st)
        outs[knob] = (yb, y, N.stats_decode(st) if st is not None else None, name)
    (yb0, y0, s0, n0), (yb1, y1, s1, n1) = outs[0], outs[2]
    assert "gemm6" not in n0 and "gemm6" in n1, (n0, n1)
    assert torch.equal(torch.isnan(yb0.float()), torch.isnan(yb1.float()))  # nothing outside the slice was written
    want = _expected(x, w, res, scale, shift, flags)
    for y in (y0, y1):
        err = (y.double() - want).abs()
        # one bf16 rounding (2^-8 relative, half an ulp), two with a residual; f32 accumulation error is far below
        bound = (2.0 ** -8 if res is None else 2.0 ** -7) * want.abs() + 2e-2
        assert bool((err <= bound).all()), float((err - bound).max())
    assert ((y1.double() - want).norm() / want.norm()).item() < 3e-3
    if s0 is not None:
        # statistics = sums of the STORED values and their squares
        v = y1.double().reshape(-1, Cout)
        torch.testing.assert_close(s1[0].double().cpu(), v.sum(0).cpu(), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(s1[1].double().cpu(), (v * v).sum(0).cpu(), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(s1, s0, rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("fm", [4, 5, 6, 7])
def test_every_tile_height(fm):
    shape = (11, 23, 19, 320, 200)  # M = 4807
    flags = N.VT_CONV_RESIDUAL
    d, x, w, res, scale, shift, ldy = _operands(shape, flags, True, seed=fm)
    yb = torch.full((*shape[:3], ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = yb[..., 32:32 + shape[4]]
    name = _run(2, d, x, w, y, scale, shift, res, None, fm=fm)
    assert name == f"gemm6_kernel<bf16,2x4+4 waves,FM{fm}>", name
    want = _expected(x, w, res, scale, shift, flags)
    assert ((y.double() - want).norm() / want.norm()).item() < 3e-3
    assert bool(torch.isnan(yb[..., :32].float()).all()) and bool(torch.isnan(yb[..., 32 + shape[4]:].float()).all())


def test_default_dispatch_takes_the_big_k_layers_only():
    """Without the knob: the concat layer of VoVNet-39's second stage (at a quarter of the batch) runs here, a 64-channel
    1x1 and a 3x3 do not."""
    d, x, w, res, scale, shift, ldy = _operands((64, 56, 56, 768, 256), 0, False, seed=1)
    y = torch.empty(64, 56, 56, 256, device="cuda", dtype=torch.bfloat16)
    assert "gemm6" in _run(1, d, x, w, y, None, None, None, None)
    d, x, w, res, scale, shift, ldy = _operands((64, 56, 56, 64, 64), 0, False, seed=2)
    y = torch.empty(64, 56, 56, 64, device="cuda", dtype=torch.bfloat16)
    assert "gemm6" not in _run(1, d, x, w, y, None, None, None, None)
