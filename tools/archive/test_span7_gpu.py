# ARCHIVED with tools/archive/vt_igemm_span7.hip (NOTEBOOK R5.14): passed 42 / 42 on the GPU when the kernel was in the build.
"""vt_igemm_span7.hip (span6 with two K-steps per tick: half the workgroup barriers per MFMA) against vt_igemm_span6.hip on
identical operands: BIT-EXACT -- same products, same summation order per accumulator, same epilogues (ConvNormAct 3x3
stride 1, reference components.py:26-35, forward and stride-1 data gradient).  span6 itself is pinned to the span kernel
bit for bit in test_span6_gpu.py and to the float64 oracle through test_kernels_gpu.py / test_fullsize_gpu.py.

Matrix: 2 / 4 / 8 chunks of 32 channels (Cin = 64, 128, 256: the 18-step period once, twice, four times per tile), 4 / 5 / 6
piece taps (map widths 14, 28 .. 45, 56), pixel counts that are not multiples of 32, odd sizes, one to three filter tiles
(3 does not divide the 32 workgroups of an XCD), a partial last filter tile (Cout = 200), channel-slice operands, several
tiles per group (the loaders run ahead across tile boundaries), every epilogue mode, the data gradient's flipped taps.  And
the shapes it must leave to span6: an odd chunk count (Cin = 96), maps too wide for the six-slot filter ring (W = 96)."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# B, Cin, Cout, H, W
SHAPES = [
    (96, 128, 128, 28, 28),   # the dominant layer's geometry at 3/8 of the batch: 5 piece taps, two tiles per group
    (40, 64, 128, 45, 37),    # two chunks (one period per tile), odd sizes, M = 66,600 (not a multiple of 32)
    (256, 256, 256, 14, 14),  # 4 piece taps, eight chunks, two filter tiles
    (24, 128, 128, 56, 56),   # 6 piece taps (Wp = 57), VoVNet-39 stage 2
    (33, 64, 256, 29, 41),    # odd batch, two filter tiles
    (12, 128, 384, 41, 52),   # three filter tiles
    (96, 192, 200, 20, 20),   # six chunks, a partial second filter tile
    (256, 64, 64, 28, 28),    # 64 output channels (forced only)
]
NOT_TAKEN = [
    (16, 96, 160, 64, 80),    # three chunks, and 160 columns (span6 splits them)
    (8, 64, 128, 96, 96),     # Wp = 97: 28 pieces per chunk slot
]
MODES = [("stats", N.VT_CONV_STATS), ("plain", 0), ("residual", N.VT_CONV_RESIDUAL),
         ("affine_relu", N.VT_CONV_AFFINE | N.VT_CONV_RELU),
         ("affine_relu_residual", N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL)]


def _desc(B, Cin, Cout, H, W, ldx, ldy, ldr, flags, flip):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, ldx
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, W, 1, 1, -1, -1
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, ldy, H, W, 1, 1
    d.ldw, d.ldr, d.flags, d.ntaps = 9 * Cin, ldr, flags, 9
    for i in range(9):
        t = 8 - i if flip else i  # flip: the tap order of a stride-1 data gradient (filter rotated by 180 degrees)
        d.dh[i], d.dw[i] = t // 3, t % 3
    return d


def _run(span7, d, x, w, y, scale, shift, res, stats):
    N.set_knob("VT_SPAN7", span7)
    N.set_knob("VT_SPAN6", 2)
    try:
        N.check(N.lib().vt_conv_igemm(C.byref(d), vp(x), vp(w), vp(y), vp(scale) if scale is not None else None,
                                      vp(shift) if shift is not None else None, vp(res) if res is not None else None,
                                      vp(stats) if stats is not None else None, stream()))
        torch.cuda.synchronize()
        return N.last_kernel_name()
    finally:
        N.set_knob("VT_SPAN7", 1)
        N.set_knob("VT_SPAN6", 1)


def _both(shape, mode):
    B, Cin, Cout, H, W = shape
    flags = mode[1]
    torch.manual_seed(sum(shape))
    slices = shape[0] % 2 == 0  # half of the shapes: operands are channel slices of wider buffers
    ldx, ldy = (Cin + 32, Cout + 64) if slices else (Cin, Cout)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin] if slices else xb
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=mode[0] == "residual")
    outs = []
    for span7 in (0, 2):
        yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
        y = yb[..., 32:32 + Cout] if slices else yb
        st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
        name = _run(span7, d, x, w, y, scale, shift, res, st)
        outs.append((yb, N.stats_decode(st) if st is not None else None, name))
    return outs


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_span7_is_bit_identical_to_span6(shape, mode):
    (y0, s0, n0), (y1, s1, n1) = _both(shape, mode)
    assert "span6" in n0 and "span7" in n1, (n0, n1)
    assert torch.equal(torch.isnan(y0.float()), torch.isnan(y1.float()))  # nothing outside the slice was written
    assert torch.equal(torch.nan_to_num(y0.float()), torch.nan_to_num(y1.float()))
    if s0 is not None:
        torch.testing.assert_close(s1, s0, rtol=1e-6, atol=1e-3)  # the same values summed in fixed point


@pytest.mark.parametrize("shape", NOT_TAKEN, ids=lambda s: "x".join(map(str, s)))
def test_shapes_left_to_span6(shape):
    (y0, s0, n0), (y1, s1, n1) = _both(shape, MODES[1])
    assert "span6" in n0 and "span6" in n1, (n0, n1)
    assert torch.equal(torch.nan_to_num(y0.float()), torch.nan_to_num(y1.float()))
