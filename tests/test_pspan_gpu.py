"""vt_igemm_pspan.hip (round 4): the persistent span kernel with the whole filter resident in LDS (eight compute + four
loader waves per CU) on the short-K convolutions of the first stages -- 3x3 stride 1 (reference components.py:26-35 in
backbones/darknet.py:23-24, vovnet.py:41-44), 3x3 stride 2 through the space-to-depth view (darknet.py:35,43), 1x1, and
the 2x2-tap depth-to-space data gradient.  Against the float64 convolution of the same bf16 operands (6e-3 relative L2:
one bf16 rounding of the output) and against the other conv kernels on the same operands (2e-3: another summation
order), for the training epilogue (raw output + BatchNorm statistics), the inference epilogue (affine + ReLU + residual
into a channel slice of a wider buffer) and the accumulating one (data gradients)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import filler
from vision_toolbox import _native as N

from gpu_util import TD, conv_desc, krsc, nhwc, rel_err, rounded, stream, to_nchw, tol, vp

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, Cout, k, s, H, W
    (2, 32, 32, 3, 1, 20, 12),     # one chunk, 32-wide tile, tiles end inside rows and images
    (2, 64, 64, 3, 1, 16, 16),     # two chunks
    (1, 32, 64, 3, 1, 33, 17),     # odd map
    (2, 32, 128, 3, 1, 12, 12),    # 128-wide tile
    (3, 32, 24, 3, 1, 9, 9),       # 24 of 32 filter columns
    (2, 128, 128, 1, 1, 12, 12),   # 1x1: one step per chunk, four chunks, three span sets in flight
    (2, 64, 40, 1, 1, 7, 5),
    (1, 32, 32, 3, 1, 130, 70),    # several tiles per workgroup range, long halo
    (2, 32, 64, 3, 2, 56, 56),     # stride 2: the first stride-2 conv of the Darknets at toy size
    (2, 64, 64, 3, 2, 64, 72),     # two chunks: every plane span is reloaded
    (3, 32, 32, 3, 2, 20, 12),
    (5, 32, 64, 3, 2, 8, 8),       # Wo = 4: most fragments wrap rows
    (2, 32, 16, 3, 2, 4, 6),       # a map smaller than a tile
    (7, 32, 64, 3, 2, 2, 2),       # Wo = 1
    (1, 32, 64, 3, 2, 224, 224),   # Wo = 112: the longest plane spans
    (1, 96, 32, 3, 1, 40, 24),     # three chunks
]


def _pad(k, s):
    return -((s - k) // 2)


def _run(case, x, w, flags, knob, scale=None, shift=None, res=None, ldy=None, coff=0):
    B, Cin, Cout, k, s, H, W_ = case
    dtype = N.VT_BF16
    pad = _pad(k, s)
    xd, wd = nhwc(x, dtype), krsc(w, dtype)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W_ + 2 * pad - k) // s + 1
    ldy = ldy or Cout
    wide = torch.full((B, Ho, Wo, ldy), float("nan"), device="cuda", dtype=TD[dtype])
    y = wide[..., coff : coff + Cout]
    stats = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
    rd = nhwc(res, dtype) if res is not None else None
    d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, ldy, flags=flags, ldr=Cout if res is not None else 0)
    N.set_knob("VT_PSPAN", knob)
    try:
        N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), vp(scale), vp(shift), vp(rd), vp(stats), stream()))
        name = N.last_kernel_name()
        torch.cuda.synchronize()
    finally:
        N.set_knob("VT_PSPAN", 1)
    return wide, y, stats, name


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_persistent_span_kernel_forward_statistics_and_epilogues(case):
    B, Cin, Cout, k, s, H, W = case
    dtype = N.VT_BF16
    pad = _pad(k, s)
    x = filler.tensor(f"psx{case}", (B, Cin, H, W))
    w = filler.tensor(f"psw{case}", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, pad)
    # training epilogue: raw output + statistics of the stored values
    _, y, stats, name = _run(case, x, w, N.VT_CONV_STATS, 2)
    assert "pspan" in name, name
    got = to_nchw(y)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < tol(dtype), name
    st = N.stats_decode(stats).cpu()
    yy = y.double().reshape(-1, Cout).cpu()
    np.testing.assert_allclose(st[0], yy.sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (yy * yy).sum(0), rtol=1e-4, atol=1e-3)
    # another kernel on the same operands: same products in another order
    _, y0, _, name0 = _run(case, x, w, N.VT_CONV_STATS, 0)
    assert "pspan" not in name0, name0
    assert rel_err(to_nchw(y), to_nchw(y0)) < 2e-3
    # inference epilogue into a channel slice of a wider buffer: relu(z * scale + shift) + residual
    sc = (filler.tensor(f"pss{case}", (Cout,)).abs() + 0.5).cuda()
    sf = (filler.tensor(f"psf{case}", (Cout,)) * 0.1).cuda()
    res = filler.tensor(f"psr{case}", tuple(ref.shape))
    ref_aff = torch.relu(ref * sc.double().cpu()[None, :, None, None] + sf.double().cpu()[None, :, None, None])
    ref_aff = ref_aff.to(torch.bfloat16).double() + rounded(res, dtype).double()
    wide, ya, _, name = _run(case, x, w, N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL, 2, sc, sf, res,
                             ldy=Cout + 24, coff=8)
    assert "pspan" in name, name
    assert rel_err(to_nchw(ya), ref_aff) < tol(dtype)
    assert torch.isnan(wide[..., :8]).all() and torch.isnan(wide[..., 8 + Cout :]).all()  # nothing outside the slice
    # accumulating epilogue (a data gradient's): y = conv + residual, no affine
    _, yr, _, name = _run(case, x, w, N.VT_CONV_RESIDUAL, 2, None, None, res)
    assert "pspan" in name, name
    assert rel_err(to_nchw(yr), ref.to(torch.bfloat16).double() + rounded(res, dtype).double()) < tol(dtype)


def test_persistent_span_kernel_leaves_the_layers_it_is_not_built_for():
    """long K (the resident filter would not fit), more than 128 output channels, odd stride-2 maps: other kernels"""
    dtype = N.VT_BF16
    N.set_knob("VT_PSPAN", 2)
    try:
        for (B, Cin, Cout, k, s, H, W) in ((2, 128, 128, 3, 1, 8, 8), (2, 32, 160, 3, 1, 8, 8), (2, 32, 64, 3, 2, 9, 8)):
            pad = _pad(k, s)
            x = filler.tensor("psnx", (B, Cin, H, W))
            w = filler.tensor("psnw", (Cout, Cin, k, k), scale=0.05)
            xd, wd = nhwc(x, dtype), krsc(w, dtype)
            Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            y = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=TD[dtype])
            d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, Cout)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), None, None, None, None, stream()))
            assert "pspan" not in N.last_kernel_name(), (Cin, Cout, H, N.last_kernel_name())
            ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, pad)
            assert rel_err(to_nchw(y), ref) < tol(dtype)
    finally:
        N.set_knob("VT_PSPAN", 1)
