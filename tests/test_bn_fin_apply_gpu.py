"""vt_bn_finalize_apply / vt_bn_bwd_finalize_apply (round 6): the BatchNorm finalize step INSIDE the streaming launch that
consumes its coefficients (the first workgroups finalize and publish with device-scope stores, every workgroup polls a
counter once) against the two launches each replaces -- vt_bn_finalize + vt_bn_act_apply (reference components.py:36-44
forward: batch statistics, running-statistics update, normalise + ReLU + DarknetBlock's add) and vt_bn_bwd_finalize +
vt_bn_act_bwd_apply (their autograd backward).  The arithmetic is the same bit for bit, so every output must be EQUAL:
coefficients, running statistics, batch counter, y / dz, d(gamma), d(beta).  Shapes cover grids smaller than the number of
finalizing workgroups (tiny M), many more workgroups than CUs (the hand-off must not depend on residency), channel counts
that are not multiples of 16, channel slices, f32 and bf16, with and without residual / ReLU."""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import TD, stream, vp

pytestmark = pytest.mark.gpu

SHAPES = [(256 * 28 * 28, 128), (256 * 14 * 14, 256), (256 * 7 * 7, 1024), (64 * 112 * 112, 32), (37, 64), (5000, 160), (1, 8), (70000, 24)]


def _stats(z):
    """a statistics buffer holding sum z, sum z^2 per channel, spread over the replicas"""
    Cc = z.shape[1]
    st = N.stats_buffer(Cc)
    zz = z.float().double()
    parts = torch.chunk(torch.arange(z.shape[0], device="cuda"), N.VT_STAT_REPLICAS)
    for r, idx in enumerate(parts):
        if idx.numel():
            N.stats_encode(st, 0, zz[idx].sum(0), r)
            N.stats_encode(st, 1, (zz[idx] ** 2).sum(0), r)
    return st


@pytest.mark.parametrize("dtype", [N.VT_BF16, N.VT_F32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_forward_finalize_inside_the_normalise_launch_is_bit_identical(shape, dtype):
    M, Cc = shape
    if dtype == N.VT_F32 and M * Cc > 30_000_000:
        pytest.skip("f32 copy of the largest tensors: covered in bf16")
    torch.manual_seed(M % 977 + Cc)
    td = TD[dtype]
    slices = Cc in (256, 160)
    ld = Cc + 16 if slices else Cc
    zb = (torch.randn(M, ld, device="cuda") * 1.7 + 0.3).to(td)
    z = zb[:, 8:8 + Cc] if slices else zb
    res = torch.randn(M, Cc, device="cuda").to(td) if Cc % 3 else None
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
    lib = N.lib()
    outs = []
    for fused in (False, True):
        st = _stats(z)
        rm, rv = torch.full((Cc,), 0.1, device="cuda"), torch.full((Cc,), 0.9, device="cuda")
        nbt = torch.full((1,), 7, dtype=torch.int64, device="cuda")
        coef = torch.zeros(4, Cc, device="cuda")
        yb = torch.full((M, ld), float("nan"), device="cuda", dtype=td)
        y = yb[:, 8:8 + Cc] if slices else yb
        before = N.launch_count()
        if fused:
            N.check(lib.vt_bn_finalize_apply(vp(st), Cc, float(M), vp(gamma), vp(beta), 1e-5, 0.1, vp(rm), vp(rv), vp(nbt), vp(coef[0]),
                                             vp(coef[1]), vp(coef[2]), vp(coef[3]), vp(z), ld, vp(res), Cc, vp(y), ld, M, 1,
                                             dtype, stream()))
        else:
            N.check(lib.vt_bn_finalize(vp(st), Cc, float(M), vp(gamma), vp(beta), 1e-5, 0.1, vp(rm), vp(rv), vp(nbt), vp(coef[0]),
                                       vp(coef[1]), vp(coef[2]), vp(coef[3]), stream()))
            N.check(lib.vt_bn_act_apply(vp(z), ld, vp(coef[0]), vp(coef[1]), vp(res), Cc, vp(y), ld, M, Cc, 1, dtype, stream()))
        torch.cuda.synchronize()
        assert N.launch_count() - before == (1 if fused else 2)
        outs.append((coef, rm, rv, nbt, yb))
    for a, b in zip(*outs):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
        assert torch.equal(torch.isnan(a.float()), torch.isnan(b.float()))
    n = C.c_uint32(0)
    N.check(lib.vt_bn_bwd_fused_timeouts(C.byref(n)))
    assert n.value == 0


@pytest.mark.parametrize("train", [1, 0], ids=["train", "frozen_stats"])
@pytest.mark.parametrize("dtype", [N.VT_BF16, N.VT_F32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_backward_finalize_inside_the_apply_launch_is_bit_identical(shape, dtype, train):
    M, Cc = shape
    if dtype == N.VT_F32 and M * Cc > 30_000_000:
        pytest.skip("f32 copy of the largest tensors: covered in bf16")
    torch.manual_seed(M % 977 + Cc + 1)
    td = TD[dtype]
    dy = torch.randn(M, Cc, device="cuda").to(td)
    z = (torch.randn(M, Cc, device="cuda") * 1.7 + 0.3).to(td)
    scale, shift = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
    mean, invstd = torch.randn(Cc, device="cuda") * 0.1 + 0.3, torch.rand(Cc, device="cuda") + 0.5
    lib = N.lib()
    sums = N.stats_buffer(Cc)
    N.check(lib.vt_bn_act_bwd_reduce(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, 1, dtype, vp(sums),
                                     stream()))
    outs = []
    for fused in (False, True):
        dg, db = torch.full((Cc,), 0.25, device="cuda"), torch.full((Cc,), -0.5, device="cuda")
        coef = torch.zeros(3, Cc, device="cuda")
        dz = torch.full((M, Cc), float("nan"), device="cuda", dtype=td)
        if fused:
            N.check(lib.vt_bn_bwd_finalize_apply(vp(sums), Cc, float(M), 1.0, vp(scale), vp(shift), vp(mean), vp(invstd), train, vp(dg),
                                                 vp(db), vp(coef), vp(dy), Cc, vp(z), Cc, vp(dz), Cc, M, 1, dtype, stream()))
        else:
            N.check(lib.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 1.0, vp(scale), vp(mean), vp(invstd), train, vp(dg), vp(db), vp(coef),
                                           stream()))
            N.check(lib.vt_bn_act_bwd_apply(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(coef), vp(dz), Cc, M, Cc, 1, dtype,
                                            stream()))
        torch.cuda.synchronize()
        outs.append((coef, dg, db, dz))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    n = C.c_uint32(0)
    N.check(lib.vt_bn_bwd_fused_timeouts(C.byref(n)))
    assert n.value == 0
