"""Parity at BASELINE.json's FULL sizes (batch 256 @224): the conv kernels the benchmark actually
runs -- 256/224-row span tiles, the stem kernel, the stride-2 general kernel, the all-taps filter
gradient -- checked where the oracle can still follow:

* forward: 96 random output pixels per layer are recomputed on the CPU in float64 from the same
  bf16-rounded operands (one dot product per pixel and channel); every stored output must also be
  consistent with the BN statistics the epilogue accumulated (a checksum over ALL 25-800 M outputs);
* filter gradient: linearity in dz -- wgrad(dz1 + dz2) = wgrad(dz1) + wgrad(dz2) -- and agreement with
  the torch CPU conv-backward on the same bf16-rounded tensors (full reduction over all pixels).

Tolerances: bf16 storage of the output (rel 2^-8 per element, 6e-3 in L2), f32 accumulation."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vision_toolbox import _native as N

from gpu_util import conv_desc, stream, vp

pytestmark = pytest.mark.gpu
B = 256

LAYERS = [  # Cin(stored), Cout, k, s, H   -- CSPDarknet-53 @224 shapes
    (128, 128, 3, 1, 28),   # dominant layer: span kernel, 224-row tiles
    (256, 256, 3, 1, 14),   # span kernel, two N tiles
    (64, 64, 1, 1, 112),    # 1x1, HBM bound, 64-wide tiles
    (8, 32, 3, 1, 224),     # stem (RGB padded to one 16-byte pixel): vt_stem.hip
    (32, 64, 3, 2, 224),    # stride 2: general gather kernel
    (512, 512, 3, 1, 7),    # small map, general kernel 128x128
]


def _rand(shape, scale, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "x".join(map(str, l)))
def test_forward_conv_at_batch_256_spot_checked_in_float64(layer):
    Cin, Cout, k, s, H = layer
    pad = -((s - k) // 2)
    x = _rand((B, H, H, Cin), 1.0, 1)
    w = _rand((Cout, k, k, Cin), (2.0 / (Cin * k * k)) ** 0.5, 2)
    Ho = (H + 2 * pad - k) // s + 1
    y = torch.full((B, Ho, Ho, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(N.VT_STAT_REPLICAS, 2, Cout, device="cuda")
    d = conv_desc(N.VT_BF16, x, Cin, Cout, k, s, pad, Cout, flags=N.VT_CONV_STATS)
    before = N.launch_count()
    N.check(N.lib().vt_conv_igemm(C.byref(d), vp(x), vp(w), vp(y), None, None, None, vp(stats), stream()))
    torch.cuda.synchronize()
    assert N.launch_count() > before
    assert torch.isfinite(y.float()).all()
    # checksum over every output: the epilogue's statistics are those of the stored values
    st = stats.double().sum(0)
    yy = y.double().reshape(-1, Cout)
    np.testing.assert_allclose(st[0].cpu(), yy.sum(0).cpu(), rtol=2e-4, atol=2e-2 * (yy.shape[0] ** 0.5))
    np.testing.assert_allclose(st[1].cpu(), (yy * yy).sum(0).cpu(), rtol=2e-4)
    # spot check in float64, including image borders and the last image
    rs = np.random.RandomState(0)
    pts = [(0, 0, 0), (B - 1, Ho - 1, Ho - 1), (B - 1, 0, Ho - 1), (17, Ho - 1, 0)]
    pts += [(int(rs.randint(B)), int(rs.randint(Ho)), int(rs.randint(Ho))) for _ in range(92)]
    wd = w.double().cpu()  # [Cout][k][k][Cin]
    got, ref = [], []
    for b, i, j in pts:
        patch = torch.zeros(k, k, Cin, dtype=torch.float64)
        for r in range(k):
            for t in range(k):
                hi, wi = i * s - pad + r, j * s - pad + t
                if 0 <= hi < H and 0 <= wi < H:
                    patch[r, t] = x[b, hi, wi].double().cpu()
        ref.append((wd * patch).sum((1, 2, 3)))
        got.append(y[b, i, j].double().cpu())
    got, ref = torch.stack(got), torch.stack(ref)
    assert ((got - ref).norm() / ref.norm()).item() < 6e-3
    assert ((got - ref).abs() <= 1e-2 * ref.abs() + 2e-2).all()


@pytest.mark.parametrize("layer", [(128, 128, 3, 1, 28), (32, 32, 3, 1, 112), (64, 128, 3, 2, 112)],
                         ids=lambda l: "x".join(map(str, l)))
def test_filter_gradient_at_batch_256_linear_and_equal_to_cpu_autograd(layer):
    Cin, Cout, k, s, H = layer
    pad = -((s - k) // 2)
    Ho = (H + 2 * pad - k) // s + 1
    x = _rand((B, H, H, Cin), 1.0, 3)
    dz1, dz2 = _rand((B, Ho, Ho, Cout), 1.0, 4), _rand((B, Ho, Ho, Cout), 1.0, 5)
    dzs = (dz1.float() + dz2.float()).to(torch.bfloat16)
    d = conv_desc(N.VT_BF16, x, Cin, Cout, k, s, pad, Cout)

    def wgrad(dz):
        dw = torch.zeros(Cout, k, k, Cin, device="cuda")
        N.check(N.lib().vt_conv_wgrad(C.byref(d), vp(x), vp(dz), vp(dw), k * k * Cin, stream()))
        torch.cuda.synchronize()
        return dw

    g1, g2, gs = wgrad(dz1), wgrad(dz2), wgrad(dzs)
    # linearity up to the bf16 rounding of (dz1 + dz2): compare against the f32 sum of the parts
    lin = ((gs - (g1 + g2)).norm() / (g1 + g2).norm()).item()
    assert lin < 4e-3, lin
    # full reduction against torch CPU autograd on the same rounded tensors (float32 accumulate)
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    dc = dz1.float().cpu().permute(0, 3, 1, 2).contiguous()
    wz = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(xc, wz, None, s, pad).backward(dc)
    ref = wz.grad.permute(0, 2, 3, 1)
    err = ((g1.cpu() - ref).norm() / ref.norm()).item()
    assert err < 2e-4, err
