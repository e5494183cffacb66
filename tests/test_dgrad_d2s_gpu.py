"""Data gradient of the 3x3 stride-2 convolutions (the first conv of every Darknet / CSPDarknet stage,
darknet.py:40-45 / :80-86; autograd backward of nn.Conv2d w.r.t. its input) as ONE depth-to-space launch
(VT_CONV_D2S): the four parity classes of d(x) become the column blocks of a 2x2-tap filter image over dz, zero taps
where a class has fewer.  Checked against (a) the four parity-class launches it replaces, on the same operands, and
(b) float64 conv_transpose2d of the same bf16 values."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

K, S, PAD = 3, 2, 1
TAPS = [(1, 1), (1, 0), (0, 1), (0, 0)]


def _class(ph, pw):
    r0, t0 = (ph + PAD) % S, (pw + PAD) % S
    rows, cols = list(range(r0, K, S)), list(range(t0, K, S))
    eh, ew = (ph + PAD - r0) // S, (pw + PAD - t0) // S
    return rows, cols, eh, ew


def _pack(lib, w, out_view, sel, Cout, Cin):
    arr = (C.c_int32 * len(sel))(*sel)
    N.check(lib.vt_pack_dgrad_filter(vp(w), N.VT_BF16, 9 * Cin, vp(out_view), N.VT_BF16, arr, len(sel), Cout, 9, Cin,
                                     stream()))


def _desc(B, Hz, Wz, Cout, ldz, H, W, ncol, ldy, ldr, ntaps, offs, flags, oh0=0, ow0=0, Hc=None, Wc=None):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, Hz, Wz, Cout, ldz
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = Hc or H // 2, Wc or W // 2, 1, 1, 0, 0
    d.Cout, d.ldy, d.oH, d.oW = ncol, ldy, H, W
    d.oHs, d.oWs, d.oh0, d.ow0 = 2, 2, oh0, ow0
    d.ldw, d.ldr, d.flags, d.ntaps = ntaps * Cout, ldr, flags, ntaps
    for i, (a, b) in enumerate(offs):
        d.dh[i], d.dw[i] = a, b
    return d


# B, Cin (channels of x), Cout (channels of dz), H, W of x, ld of dx, residual
CASES = [
    (4, 32, 64, 112, 112, 32, False),   # stage 0 of the Darknets at 1/64 of the batch: the span kernel's 128-column tile
    (2, 16, 24, 20, 28, 16, True),      # small: general kernel, folded addend
    (3, 32, 48, 36, 52, 64, True),      # d(x) is a channel slice of a wider buffer
    (2, 64, 64, 56, 56, 64, False),     # 256 columns
    (2, 32, 64, 24, 40, 48, True),      # 128 columns, folded addend, channel slice: the persistent span kernel when forced
]


@pytest.mark.parametrize("pspan", [1, 2], ids=["default_dispatch", "persistent_span_forced"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ldx,with_res", CASES)
def test_depth_to_space_dgrad_equals_the_four_class_launches(B, Cin, Cout, H, W, ldx, with_res, pspan):
    torch.manual_seed(Cin + H)
    dev = "cuda"
    lib = N.lib()
    Hz, Wz = H // 2, W // 2
    dz = torch.randn(B, Hz, Wz, Cout, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device=dev) * 0.1).to(torch.bfloat16)  # [n][tap][c]
    res = torch.randn(B, H, W, ldx, device=dev).to(torch.bfloat16) if with_res else None
    rflag = N.VT_CONV_RESIDUAL if with_res else 0
    # (a) four class launches
    dx_ref = torch.zeros(B, H, W, ldx, device=dev, dtype=torch.bfloat16)
    for ph in range(2):
        for pw in range(2):
            rows, cols, eh, ew = _class(ph, pw)
            sel = [r * K + t for r in rows for t in cols]
            offs = [(eh - u, ew - v) for u in range(len(rows)) for v in range(len(cols))]
            wd = torch.empty(Cin, len(sel), Cout, device=dev, dtype=torch.bfloat16)
            _pack(lib, w, wd, sel, Cout, Cin)
            d = _desc(B, Hz, Wz, Cout, Cout, H, W, Cin, ldx, ldx, len(sel), offs, rflag, ph, pw)
            N.check(lib.vt_conv_igemm(C.byref(d), vp(dz), vp(wd), vp(dx_ref), None, None, vp(res), None, stream()))
    # (b) one depth-to-space launch
    dx = torch.zeros(B, H, W, ldx, device=dev, dtype=torch.bfloat16)
    wd4 = torch.empty(4, Cin, 4, Cout, device=dev, dtype=torch.bfloat16)
    for ph in range(2):
        for pw in range(2):
            rows, cols, eh, ew = _class(ph, pw)
            sel = []
            for (a, b) in TAPS:
                u, v = eh - a, ew - b
                sel.append(rows[u] * K + cols[v] if 0 <= u < len(rows) and 0 <= v < len(cols) else -1)
            _pack(lib, w, wd4[2 * ph + pw], sel, Cout, Cin)
    d = _desc(B, Hz, Wz, Cout, Cout, H, W, 4 * Cin, ldx, ldx, 4, TAPS, N.VT_CONV_D2S | rflag)
    N.set_knob("VT_PSPAN", pspan)  # (2: vt_igemm_pspan.hip wherever it applies -- <= 128 columns, dz channels a multiple of 32)
    try:
        N.check(lib.vt_conv_igemm(C.byref(d), vp(dz), vp(wd4), vp(dx), None, None, vp(res), None, stream()))
        name = N.last_kernel_name()
    finally:
        N.set_knob("VT_PSPAN", 1)
    if pspan == 2 and 4 * Cin <= 128 and Cout % 32 == 0:
        assert "pspan" in name, name
    torch.cuda.synchronize()
    a, b = dx[..., :Cin].float(), dx_ref[..., :Cin].float()
    # same products; the tile shapes of the two launches may order the K chunks differently: one bf16 rounding
    assert ((a - b).abs() <= 2.0 ** -7 * b.abs().clamp_min(2.0 ** -6)).all()
    if ldx > Cin:
        assert (dx[..., Cin:] == 0).all()  # the rest of the wider buffer is untouched
    # float64 on the same bf16 operands
    w_oihw = w.double().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(dz.double().permute(0, 3, 1, 2), w_oihw, stride=2, padding=1, output_padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if with_res:
        ref = ref + res[..., :Cin].double()
    err = (a.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 6e-3, err


def test_depth_to_space_rejects_a_strided_epilogue_it_cannot_honour():
    lib = N.lib()
    t = torch.zeros(4096, device="cuda", dtype=torch.bfloat16)
    d = _desc(1, 4, 4, 8, 8, 8, 8, 32, 8, 8, 4, TAPS, N.VT_CONV_D2S | N.VT_CONV_RELU)
    assert lib.vt_conv_igemm(C.byref(d), vp(t), vp(t), vp(t), None, None, None, None, stream()) == N.VT_ERR_UNSUPPORTED
    d = _desc(1, 4, 4, 8, 8, 8, 8, 32, 8, 8, 4, TAPS, N.VT_CONV_D2S, oh0=1)
    assert lib.vt_conv_igemm(C.byref(d), vp(t), vp(t), vp(t), None, None, None, None, stream()) != N.VT_OK
