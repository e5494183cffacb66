"""vt_stem_bwd.hip: backward of the stem unit Conv3x3(3 -> 32, s1) -> BatchNorm2d -> ReLU (darknet.py:75,
components.py:26-44) as ONE streaming pass + a combine kernel, against float64 torch on the same bf16 operands:

    g = dy * [z*scale + shift > 0],  sums = (sum g, sum g*(z - mean)*invstd)          (the BatchNorm-backward reduction)
    dW[n][t][c] = sum_p (a_n g - b_n z + d_n)[p][n] * x[p + t][c]                     (filter gradient of that dz)

The fused path never rounds dz to bf16, so the comparison is against the exact (float64) filter gradient of the
float64 dz; tolerances are f32-accumulation sized.  Shapes: odd maps (padded coordinates (H+1) x (W+1)), one and many
workgroups, the full 224-wide map, channel-slice operands (ld > C)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# B, H, W, lddy, ldz, relu
CASES = [
    (3, 13, 17, 32, 32, 1),      # a single workgroup, odd sizes
    (8, 56, 40, 32, 32, 1),      # several workgroups, chunk boundaries inside images
    (5, 31, 63, 48, 64, 1),      # operands that are channel slices of wider buffers
    (4, 224, 224, 32, 32, 1),    # the real map: halo 256 rows, ring of 1024
    (2, 40, 300, 32, 32, 0),     # wide map, no ReLU mask
    (1, 5, 824, 32, 32, 1),      # the widest map the y form takes (engine.py: one more step of halo in the 2048-row ring)
]


def _reference(x, dy, z, sc, sf, mu, istd, coef, relu):
    """float64 on the bf16 values: sums [2][C] and dW [C][9][3]"""
    xd, dyd, zd = x.double(), dy.double(), z.double()
    on = (torch.addcmul(sf, z.float(), sc) > 0) if relu else torch.ones_like(z, dtype=torch.bool)  # f32 fma, as the kernels
    g = torch.where(on, dyd, torch.zeros_like(dyd))
    s1 = g.sum((0, 1, 2))
    s2 = (g * (zd - mu.double())).sum((0, 1, 2)) * istd.double()
    a, b, d = (coef[i].double() for i in range(3))
    dz = a * g - b * zd + d
    w = torch.zeros(32, 3, 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    y = F.conv2d(xd[..., :3].permute(0, 3, 1, 2), w, padding=1)
    y.backward(dz.permute(0, 3, 1, 2))
    dw = w.grad.permute(0, 2, 3, 1).reshape(32, 9, 3)  # [n][tap][c]
    return torch.stack([s1, s2]), dw


@pytest.mark.parametrize("fixed", [0, 1], ids=["f32_atomics", "fixed_point"])
@pytest.mark.parametrize("B,H,W,lddy,ldz,relu", CASES)
def test_stem_backward_matches_float64(B, H, W, lddy, ldz, relu, fixed):
    torch.manual_seed(B * 1000 + W)
    dev = "cuda"
    Cc = 32
    x = torch.zeros(B, H, W, 8, device=dev, dtype=torch.bfloat16)
    x[..., :3] = torch.randn(B, H, W, 3, device=dev).to(torch.bfloat16)
    dyw = torch.randn(B, H, W, lddy, device=dev).to(torch.bfloat16)
    zw = (torch.randn(B, H, W, ldz, device=dev) * 1.5 + 0.4).to(torch.bfloat16)
    off_dy, off_z = (lddy - Cc), (ldz - Cc) // 2 // 8 * 8
    dy, z = dyw[..., off_dy:off_dy + Cc], zw[..., off_z:off_z + Cc]
    sc = torch.randn(Cc, device=dev) * 0.8
    sf = torch.randn(Cc, device=dev) * 0.5
    mu = torch.randn(Cc, device=dev) * 0.3 + 0.4
    istd = torch.rand(Cc, device=dev) + 0.5
    coef = torch.randn(3, Cc, device=dev) * torch.tensor([[1.0], [0.05], [0.02]], device=dev)
    sums = N.stats_buffer(Cc)
    lib = N.lib()
    gzx = torch.zeros(lib.vt_stem_bn_bwd_scratch_bytes(Cc) // 4, device=dev)
    dw = torch.full((Cc, 9, 3), 0.25, device=dev)  # the combine kernel ACCUMULATES
    st = stream()
    N.check(lib.vt_stem_bn_bwd_reduce(N.VT_BF16, B, H, W, Cc, vp(x), vp(dy), lddy, vp(z), ldz, vp(sc), vp(sf), vp(mu),
                                      vp(istd), relu, vp(sums), vp(gzx), fixed, st))
    N.check(lib.vt_stem_bn_bwd_combine(Cc, 3, vp(gzx), vp(coef), vp(dw), fixed, st))
    torch.cuda.synchronize()
    ref_s, ref_dw = _reference(x, dy, z, sc, sf, mu, istd, coef, relu)
    got_s = N.stats_decode(sums)
    scale_s = ref_s.abs().max(dim=1, keepdim=True).values
    assert ((got_s - ref_s).abs() / scale_s).max().item() < 2e-5
    got_dw = dw.double() - 0.25
    err = (got_dw - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("fixed", [0, 1], ids=["f32_atomics", "fixed_point"])
@pytest.mark.parametrize("B,H,W,lddy,ldz,relu", CASES)
def test_stem_backward_from_the_units_output(B, H, W, lddy, ldz, relu, fixed):
    """`fixed` bit 1: the pass reads y = relu(z*scale + shift) (bf16) instead of z, which the forward pass never stored.
    Mask = y > 0; z is linear in the 3x3 patch, so sum g*xhat (vt_stem_bn_bwd_s2: from G = sum g x(p+t) and W) and the b*Z
    term of dW (Z = W P with P the patch / tap-shifted-x correlations) use the EXACT z = W * patch: the float64 reference
    takes z from a float64 conv of the bf16 operands.  Nothing is recovered from y: a scale of exactly 0 is as exact."""
    torch.manual_seed(B * 1000 + W + 7)
    dev = "cuda"
    Cc = 32
    x = torch.zeros(B, H, W, 8, device=dev, dtype=torch.bfloat16)
    x[..., :3] = torch.randn(B, H, W, 3, device=dev).to(torch.bfloat16)
    wq = torch.zeros(Cc, 9, 8, device=dev, dtype=torch.bfloat16)  # the forward conv's filter image
    wq[..., :3] = (torch.randn(Cc, 9, 3, device=dev) * 0.3).to(torch.bfloat16)
    z_true = F.conv2d(x[..., :3].double().permute(0, 3, 1, 2),
                      wq[..., :3].double().reshape(Cc, 3, 3, 3).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    sc = torch.randn(Cc, device=dev) * 0.8
    sc[3] = 0.0  # a BatchNorm weight of exactly 0: y = relu(shift) says nothing about z
    sf = torch.randn(Cc, device=dev) * 0.5
    sf[3] = 0.7
    yv = z_true.float() * sc + sf
    yw = torch.randn(B, H, W, ldz, device=dev).to(torch.bfloat16)
    off_dy, off_y = (lddy - Cc), (ldz - Cc) // 2 // 8 * 8
    yw[..., off_y:off_y + Cc] = (yv.clamp_min(0) if relu else yv).to(torch.bfloat16)
    dyw = torch.randn(B, H, W, lddy, device=dev).to(torch.bfloat16)
    dy, y = dyw[..., off_dy:off_dy + Cc], yw[..., off_y:off_y + Cc]
    mu = torch.randn(Cc, device=dev) * 0.3
    istd = torch.rand(Cc, device=dev) + 0.5
    coef = torch.randn(3, Cc, device=dev) * torch.tensor([[1.0], [0.05], [0.02]], device=dev)
    sums = N.stats_buffer(Cc)
    lib = N.lib()
    gzx = torch.zeros(lib.vt_stem_bn_bwd_scratch_bytes(Cc) // 4, device=dev)
    dw = torch.full((Cc, 9, 3), 0.25, device=dev)
    st = stream()
    N.check(lib.vt_stem_bn_bwd_reduce(N.VT_BF16, B, H, W, Cc, vp(x), vp(dy), lddy, vp(y), ldz, vp(sc), vp(sf), vp(mu),
                                      vp(istd), relu, vp(sums), vp(gzx), fixed | 2, st))
    assert (N.stats_decode(sums)[1] == 0).all()  # left to vt_stem_bn_bwd_s2
    N.check(lib.vt_stem_bn_bwd_s2(Cc, vp(gzx), vp(wq), vp(mu), vp(istd), vp(sums), fixed, st))
    N.check(lib.vt_stem_bn_bwd_combine_y(Cc, 3, vp(gzx), vp(coef), vp(wq), vp(dw), fixed, st))
    torch.cuda.synchronize()
    on = (y.float() > 0) if relu else torch.ones_like(y, dtype=torch.bool)
    g = torch.where(on, dy.double(), torch.zeros_like(dy, dtype=torch.float64))
    ref_s = torch.stack([g.sum((0, 1, 2)), (g * (z_true - mu.double())).sum((0, 1, 2)) * istd.double()])
    a, b, d = (coef[i].double() for i in range(3))
    dz = a * g - b * z_true + d
    w = torch.zeros(32, 3, 3, 3, dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(x[..., :3].double().permute(0, 3, 1, 2), w, padding=1).backward(dz.permute(0, 3, 1, 2))
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(32, 9, 3)
    got_s = N.stats_decode(sums)
    scale_s = ref_s.abs().max(dim=1, keepdim=True).values
    assert ((got_s - ref_s).abs() / scale_s).max().item() < 2e-5
    err = ((dw.double() - 0.25) - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    assert err < 2e-5, err


def test_stem_backward_rejects_what_it_has_no_kernel_for():
    lib = N.lib()
    t = torch.zeros(64, device="cuda")
    assert lib.vt_stem_bn_bwd_reduce(N.VT_BF16, 1, 8, 8, 64, vp(t), vp(t), 64, vp(t), 64, vp(t), vp(t), vp(t), vp(t), 1,
                                     vp(t), vp(t), 0, stream()) == N.VT_ERR_UNSUPPORTED
    assert lib.vt_stem_bn_bwd_reduce(N.VT_F32, 1, 8, 8, 32, vp(t), vp(t), 32, vp(t), 32, vp(t), vp(t), vp(t), vp(t), 1,
                                     vp(t), vp(t), 0, stream()) == N.VT_ERR_UNSUPPORTED
