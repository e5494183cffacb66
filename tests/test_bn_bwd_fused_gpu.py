"""vt_bn_act_bwd_fused (round 6): the BatchNorm2d + ReLU backward of a ConvNormAct unit (autograd backward of reference
components.py:36-44) as ONE launch -- reduction, finalize and apply with the operands held in registers between the passes
and two device-scope grid barriers -- against the three launches it replaces (vt_bn_act_bwd_reduce, vt_bn_bwd_finalize,
vt_bn_act_bwd_apply: themselves pinned to the oracle in test_kernels_gpu.py) and against float64 on the stored values.

  * the sums are the same f32 terms in another order: 1e-5 of a column's absolute sum; d(gamma), d(beta), the three
    coefficient rows follow (1e-5 relative to their scale); dz is then equal up to the last bf16 bit of a few elements;
  * one launch where the operands fit the register file (256 channels @14x14 and smaller at batch 256), exactly three
    otherwise; channel-slice operands; 160 channels (20 chunks per row: 500 of 512 threads active); with / without the
    ReLU mask; train = 0 (running statistics: constants in backward);
  * run twice on the same inputs: bit-identical (the sums are fixed point from the workgroup level up);
  * no barrier ever ran into its wall-clock bound (vt_bn_bwd_fused_timeouts)."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# M, C, fused?
SHAPES = [
    (256 * 14 * 14, 256, True),   # DarknetBlock units of stage 3 (13 rows per thread)
    (256 * 7 * 7, 512, True),     # stage 4
    (256 * 7 * 7, 1024, True),    # the last CSP out_conv
    (128 * 14 * 14, 256, True),   # the data-parallel per-GPU batch
    (3001, 160, True),            # ragged rows, 20 chunks per row
    (77, 64, True),               # fewer rows than one workgroup pass
    (256 * 28 * 28, 128, False),  # 25 rows per thread: not offered, three launches
]


def _three(dy, z, scale, shift, mean, invstd, M, Cc, relu, train, lds):
    lib = N.lib()
    sums = N.stats_buffer(Cc)
    dgamma, dbeta = torch.full((Cc,), 0.25, device="cuda"), torch.full((Cc,), -0.5, device="cuda")
    coef = torch.empty(3, Cc, device="cuda")
    dzb = torch.full((M, lds[2]), float("nan"), device="cuda", dtype=torch.bfloat16)
    dz = dzb[:, 8:8 + Cc] if lds[2] > Cc else dzb
    N.check(lib.vt_bn_act_bwd_reduce(vp(dy), lds[0], vp(z), lds[1], vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, relu,
                                     N.VT_BF16, vp(sums), stream()))
    N.check(lib.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 1.0, vp(scale), vp(mean), vp(invstd), train, vp(dgamma), vp(dbeta),
                                   vp(coef), stream()))
    N.check(lib.vt_bn_act_bwd_apply(vp(dy), lds[0], vp(z), lds[1], vp(scale), vp(shift), vp(coef), vp(dz), lds[2], M, Cc, relu,
                                    N.VT_BF16, stream()))
    torch.cuda.synchronize()
    return N.stats_decode(sums), dgamma, dbeta, coef, dzb


def _fused(dy, z, scale, shift, mean, invstd, M, Cc, relu, train, lds):
    lib = N.lib()
    sums = N.stats_buffer(Cc)
    sync = torch.zeros(4, dtype=torch.int32, device="cuda")
    dgamma, dbeta = torch.full((Cc,), 0.25, device="cuda"), torch.full((Cc,), -0.5, device="cuda")
    coef = torch.empty(3, Cc, device="cuda")
    dzb = torch.full((M, lds[2]), float("nan"), device="cuda", dtype=torch.bfloat16)
    dz = dzb[:, 8:8 + Cc] if lds[2] > Cc else dzb
    before = N.launch_count()
    N.check(lib.vt_bn_act_bwd_fused(vp(dy), lds[0], vp(z), lds[1], vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, relu,
                                    N.VT_BF16, float(M), 1.0, train, vp(sums), vp(sync), vp(dgamma), vp(dbeta), vp(coef), vp(dz),
                                    lds[2], stream()))
    torch.cuda.synchronize()
    return (N.stats_decode(sums), dgamma, dbeta, coef, dzb), N.launch_count() - before, sync.cpu().tolist()


@pytest.mark.parametrize("relu,train", [(1, 1), (0, 1), (1, 0)], ids=["relu", "no_act", "frozen_stats"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_one_launch_equals_the_three_launches(shape, relu, train):
    M, Cc, fused = shape
    torch.manual_seed(M % 1000 + Cc)
    slices = Cc in (256, 160)
    lds = (Cc + 16, Cc + 32, Cc + 24) if slices else (Cc, Cc, Cc)
    dyb = torch.randn(M, lds[0], device="cuda").to(torch.bfloat16)
    zb = (torch.randn(M, lds[1], device="cuda") * 1.3 + 0.4).to(torch.bfloat16)
    dy = dyb[:, 8:8 + Cc] if slices else dyb
    z = zb[:, 16:16 + Cc] if slices else zb
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.3
    mean = z.float().mean(0)
    invstd = 1.0 / torch.sqrt(z.float().var(0, unbiased=False) + 1e-5)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    ref = _three(dy, z, scale, shift, mean, invstd, M, Cc, relu, train, lds)
    got, launches, sync = _fused(dy, z, scale, shift, mean, invstd, M, Cc, relu, train, lds)
    again, _, _ = _fused(dy, z, scale, shift, mean, invstd, M, Cc, relu, train, lds)
    assert launches == (1 if fused else 3), (launches, N.last_kernel_name())
    assert sync[2] == 0, sync
    if fused:
        assert sync[0] == sync[1] and sync[0] >= 1  # every workgroup passed both barriers
    n = C.c_uint32(0)
    N.check(N.lib().vt_bn_bwd_fused_timeouts(C.byref(n)))
    assert n.value == 0
    # float64 column scales
    mask = (torch.addcmul(shift, z.float(), scale) > 0).double() if relu else torch.ones_like(z, dtype=torch.float64)
    g = dy.double() * mask
    col = torch.stack([g.abs().sum(0), (g * (z.double() - mean.double())).abs().sum(0) * invstd.double()])
    ref64 = torch.stack([g.sum(0), (g * (z.double() - mean.double())).sum(0) * invstd.double()])
    assert ((got[0] - ref[0]).abs() / col).max().item() < 1e-5
    assert ((got[0] - ref64).abs() / col).max().item() < 2e-5
    for k in (1, 2):  # d(gamma), d(beta): the prior content + the sums
        torch.testing.assert_close(got[k], ref[k], rtol=1e-5, atol=1e-5 * col.max().item())
    torch.testing.assert_close(got[3], ref[3], rtol=2e-5, atol=1e-6)
    a, b = ref[4], got[4]
    assert torch.equal(torch.isnan(a.float()), torch.isnan(b.float()))  # nothing outside the slice was written
    a, b = torch.nan_to_num(a.float()), torch.nan_to_num(b.float())
    torch.testing.assert_close(b, a, rtol=2.0 ** -7, atol=1e-3)
    assert (a != b).float().mean().item() < 2e-3  # a last-bit difference here and there, no more
    # run to run: bit-identical
    assert torch.equal(torch.nan_to_num(again[4].float()), b) and torch.equal(again[3], got[3]) and torch.equal(again[1], got[1])
