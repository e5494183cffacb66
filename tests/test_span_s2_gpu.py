"""The input-span kernel on the space-to-depth view of a 3x3 stride-2 convolution (vt_igemm_span.hip, flag S2; round 4):
the stride-2 ConvNormAct that opens every Darknet / CSPDarknet stage (reference backbones/darknet.py:35,43,
components.py:26-35 with stride=2, padding=ceil((3-2)/2)=1).  Against the float64 convolution of the same bf16 operands
(tolerance 6e-3 relative L2: one bf16 rounding of the output) and against the gather kernel (vt_igemm.hip) on the same
operands (same products, another summation order: 2e-3), for the training epilogue (raw output + BatchNorm statistics),
the inference epilogue (affine + ReLU + residual into a channel slice) and the plain one."""
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
    # B, Cin, Cout, H, W
    (2, 32, 64, 56, 56),     # the first stride-2 conv of the Darknets at toy size: one chunk, 64-wide tile
    (2, 64, 128, 64, 72),    # two chunks, 128-wide tile, Wo = 36
    (3, 32, 32, 20, 12),     # 32-wide tile; tiles end inside rows and images
    (1, 64, 160, 30, 34),    # two filter-column tiles, the second a quarter full
    (2, 128, 256, 28, 28),   # four chunks: every plane slot is reloaded three times
    (2, 96, 64, 16, 16),     # three chunks
    (5, 32, 64, 8, 8),       # Wo = 4: most fragments wrap rows
    (2, 32, 16, 4, 6),       # a map smaller than a tile, 16 output channels
    (7, 32, 64, 2, 2),       # Wo = 1: every position is the first of its row
    (1, 32, 64, 224, 224),   # Wo = 112: the longest spans (24 pieces)
]


def _run(x, w, Cin, Cout, flags, knob, scale=None, shift=None, res=None, ldy=None, coff=0):
    dtype = N.VT_BF16
    B, H, W_ = x.shape[0], x.shape[2], x.shape[3]
    xd, wd = nhwc(x, dtype), krsc(w, dtype)
    Ho, Wo = H // 2, W_ // 2
    ldy = ldy or Cout
    wide = torch.full((B, Ho, Wo, ldy), float("nan"), device="cuda", dtype=TD[dtype])
    y = wide[..., coff : coff + Cout]
    stats = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
    rd = nhwc(res, dtype) if res is not None else None
    d = conv_desc(dtype, xd, Cin, Cout, 3, 2, 1, ldy, flags=flags, ldr=Cout if res is not None else 0)
    N.set_knob("VT_SPAN_S2", knob)
    try:
        N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), vp(scale), vp(shift), vp(rd), vp(stats), stream()))
        name = N.last_kernel_name()
        torch.cuda.synchronize()
    finally:
        N.set_knob("VT_SPAN_S2", 0)
    return wide, y, stats, name


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_stride2_span_kernel_forward_statistics_and_fused_epilogue(case):
    B, Cin, Cout, H, W = case
    dtype = N.VT_BF16
    x = filler.tensor(f"s2x{case}", (B, Cin, H, W))
    w = filler.tensor(f"s2w{case}", (Cout, Cin, 3, 3), scale=(2.0 / (Cin * 9)) ** 0.5)
    ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, 2, 1)
    # training epilogue: raw output + statistics of the stored values
    _, y, stats, name = _run(x, w, Cin, Cout, N.VT_CONV_STATS, 2)
    assert "s2d" in name, name
    got = to_nchw(y)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < tol(dtype), name
    st = N.stats_decode(stats).cpu()
    yy = y.double().reshape(-1, Cout).cpu()
    np.testing.assert_allclose(st[0], yy.sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (yy * yy).sum(0), rtol=1e-4, atol=1e-3)
    # the gather kernel on the same operands: same products in another order
    _, y0, _, name0 = _run(x, w, Cin, Cout, N.VT_CONV_STATS, 0)
    assert "igemm_kernel" in name0, name0
    assert rel_err(to_nchw(y), to_nchw(y0)) < 2e-3
    # inference epilogue into a channel slice of a wider buffer: relu(z * scale + shift) + residual
    sc = (filler.tensor(f"s2s{case}", (Cout,)).abs() + 0.5).cuda()
    sf = (filler.tensor(f"s2f{case}", (Cout,)) * 0.1).cuda()
    res = filler.tensor(f"s2r{case}", (B, Cout, H // 2, W // 2))
    ref_aff = torch.relu(ref * sc.double().cpu()[None, :, None, None] + sf.double().cpu()[None, :, None, None])
    ref_aff = ref_aff.to(torch.bfloat16).double() + rounded(res, dtype).double()
    wide, ya, _, name = _run(x, w, Cin, Cout, N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL, 2, sc, sf, res,
                             ldy=Cout + 24, coff=8)
    assert "s2d" in name, name
    assert rel_err(to_nchw(ya), ref_aff) < tol(dtype)
    assert torch.isnan(wide[..., :8]).all() and torch.isnan(wide[..., 8 + Cout :]).all()  # nothing outside the slice
