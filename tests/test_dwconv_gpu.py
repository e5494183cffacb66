"""vt_dwconv_fwd / vt_dwconv_dgrad / vt_dwconv_wgrad (round 6, vt_dwconv.hip) through the C-ABI against torch's depthwise
convolution in float64 on the same (storage-rounded) operands -- nn.Conv2d(C, C, k, stride, padding, dilation, groups=C,
bias=False) and its autograd backward (reference components.py:26-35 with `groups = in_channels`).

The module-level cases (test_modules_gpu.py) are small; here: channel counts beyond one thread row (more 16-byte chunks per
pixel than the 256 threads of a workgroup), channel-slice operands (pixel stride > C), 5x5 / 7x7 filters (two / six tap passes
of the filter gradient), strides 2 and 3 with dilation, the residual operand of the data gradient, the batch statistics of the
forward (fixed point: compared exactly against the sums of the stored outputs in float64 at 1e-6 of their scale)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vision_toolbox import _native as N

from gpu_util import TD, stream, vp

pytestmark = pytest.mark.gpu

# B, C, H, W, k, s, pad, dil
CASES = [
    (4, 64, 19, 23, 3, 1, 1, 1),
    (3, 32, 20, 18, 3, 2, 1, 1),
    (2, 2304, 6, 5, 3, 1, 1, 1),     # 288 (bf16) / 576 (f32) chunks per pixel: more than one pass of the 256 threads
    (2, 160, 17, 17, 5, 1, 2, 1),    # 20 chunks: not a power of two (no in-wave fold)
    (2, 48, 31, 29, 7, 3, 3, 1),     # 49 taps: six passes of the filter gradient
    (3, 24, 22, 26, 3, 2, 1, 2),     # dilation: some input pixels are read by no output
    (5, 8, 9, 9, 1, 1, 0, 1),
]


@pytest.mark.parametrize("dtype", [N.VT_F32, N.VT_BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_depthwise_kernels_match_torch_float64(case, dtype):
    B, Cc, H, W, k, s, pad, dil = case
    torch.manual_seed(Cc + H + k)
    td = TD[dtype]
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // s + 1, (W + 2 * pad - dil * (k - 1) - 1) // s + 1
    slices = Cc % 16 == 0 and Cc <= 256
    ldx, ldz = (Cc + 16, Cc + 32) if slices else (Cc, Cc)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(td)
    x = xb[..., 8:8 + Cc] if slices else xb
    w = (torch.randn(Cc, k * k, device="cuda") * (1.0 / k)).contiguous()
    w_used = w.to(td).float()  # (bf16 launches round the filter)
    lib = N.lib()
    geo = (B, H, W, Cc, k, s, pad, dil, dtype)
    # ---- forward (+ statistics)
    zb = torch.full((B, Ho, Wo, ldz), float("nan"), device="cuda", dtype=td)
    z = zb[..., 16:16 + Cc] if slices else zb
    st = N.stats_buffer(Cc)
    N.check(lib.vt_dwconv_fwd(vp(x), ldx, vp(w), vp(z), ldz, vp(st), *geo, stream()))
    torch.cuda.synchronize()
    x64 = x.double().permute(0, 3, 1, 2)
    w64 = w_used.double().view(Cc, 1, k, k)
    ref = F.conv2d(x64, w64, None, s, pad, dil, groups=Cc).permute(0, 2, 3, 1)
    tol = 2e-6 if dtype == N.VT_F32 else 6e-3
    assert ((z.double() - ref).norm() / ref.norm()).item() < tol
    if slices:
        assert torch.isnan(zb[..., :16].float()).all() and torch.isnan(zb[..., 16 + Cc:].float()).all()
    got = N.stats_decode(st)
    zs = z.double().reshape(-1, Cc)
    want = torch.stack([zs.sum(0), (zs * zs).sum(0)])
    scale = torch.stack([zs.abs().sum(0), (zs * zs).sum(0)]).clamp_min(1e-30)
    assert ((got - want).abs() / scale).max().item() < 1e-6
    # ---- data gradient (+ residual) and filter gradient
    dz = torch.randn(B, Ho, Wo, Cc, device="cuda").to(td)
    res = torch.randn(B, H, W, Cc, device="cuda").to(td)
    xr = x64.clone().requires_grad_(True)
    wr = w64.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, s, pad, dil, groups=Cc).backward(dz.double().permute(0, 3, 1, 2))
    dx = torch.full((B, H, W, Cc), float("nan"), device="cuda", dtype=td)
    N.check(lib.vt_dwconv_dgrad(vp(dz), Cc, vp(w), vp(dx), Cc, None, 0, *geo, stream()))
    dxr = torch.full((B, H, W, Cc), float("nan"), device="cuda", dtype=td)
    N.check(lib.vt_dwconv_dgrad(vp(dz), Cc, vp(w), vp(dxr), Cc, vp(res), Cc, *geo, stream()))
    dw = torch.full((Cc, k * k), 0.5, device="cuda")
    N.check(lib.vt_dwconv_wgrad(vp(x), ldx, vp(dz), Cc, vp(dw), *geo, stream()))
    torch.cuda.synchronize()
    gx = xr.grad.permute(0, 2, 3, 1)
    assert ((dx.double() - gx).norm() / gx.norm()).item() < tol
    assert ((dxr.double() - (gx + res.double())).norm() / (gx + res.double()).norm()).item() < 2 * tol
    gw = wr.grad.view(Cc, k * k)
    assert (((dw.double() - 0.5) - gw).norm() / gw.norm()).item() < (2e-5 if dtype == N.VT_F32 else 1e-5)  # (f32 sums of exact products)


def test_depthwise_arguments_are_checked():
    lib = N.lib()
    x = torch.zeros(1, 4, 4, 8, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(8, 9, device="cuda")
    z = torch.zeros(1, 4, 4, 8, device="cuda", dtype=torch.bfloat16)
    assert lib.vt_dwconv_fwd(vp(x), 8, vp(w), vp(z), 8, None, 1, 4, 4, 8, 9, 1, 1, 1, N.VT_BF16, stream()) == N.VT_ERR_INVALID  # k > 7
    assert lib.vt_dwconv_fwd(vp(x), 8, vp(w), vp(z), 8, None, 1, 4, 4, 4, 3, 1, 1, 1, N.VT_BF16, stream()) == N.VT_ERR_UNSUPPORTED  # C % 8
    assert lib.vt_dwconv_fwd(vp(x), 8, vp(w), None, 8, None, 1, 4, 4, 8, 3, 1, 1, 1, N.VT_BF16, stream()) == N.VT_ERR_INVALID
    assert "vt_dwconv_fwd" in N.last_error()
    buf = torch.zeros(16, device="cuda")  # (the launch-count guard of the GPU modules wants a launch)
    N.check(lib.vt_memset(buf.data_ptr(), 0, 64, stream()))
