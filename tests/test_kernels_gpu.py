"""Kernel-level parity: every C-ABI entry point against the CPU oracle's arithmetic
(torch CPU fp32/fp64, the reference's own third-party arithmetic) on seeded inputs.

Tolerances (relative L2 unless stated): f32 path 2e-5 (exact-f32 MFMA, only the
summation order differs from ATen); bf16 path 6e-3 against a reference computed from
the SAME bf16-rounded inputs (remaining error = one bf16 rounding of the output).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import filler
from vision_toolbox import _native as N

from gpu_util import DNAME, DTYPES, TD, conv_desc, krsc, nhwc, rel_err, rounded, stream, to_nchw, tol, vp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lib_loaded():
    N.lib()
    before = N.launch_count()
    yield
    torch.cuda.synchronize()
    assert N.launch_count() > before, "no libvt_amd launch happened: the HIP path did not run"


CONV_CASES = [
    # B, Cin, Cout, k, s, H, W
    (2, 16, 32, 1, 1, 8, 8),
    (2, 16, 16, 3, 1, 9, 9),
    (2, 8, 24, 3, 2, 10, 10),
    (2, 8, 16, 6, 2, 12, 12),
    (3, 40, 72, 3, 1, 7, 5),
    (1, 64, 136, 3, 1, 14, 14),
    (2, 160, 160, 3, 1, 7, 7),
    (4, 32, 32, 3, 1, 33, 17),
    (2, 128, 64, 1, 1, 12, 12),
    (2, 256, 128, 3, 2, 14, 14),
    (1, 8, 8, 3, 1, 3, 3),
    (2, 8, 32, 3, 1, 20, 13),  # RGB-stem shape class (one 16-byte pixel): vt_stem.hip in bf16
    (1, 8, 64, 3, 1, 9, 40),
    (3, 8, 24, 3, 1, 17, 17),
    # 3x3 stride 2 on even maps, Wo = 28 / 36 / 29 / 32, one and two channel tiles: with lowered thresholds (see
    # test_stride2_all_taps_filter_gradient_in_subprocess) the bf16 filter gradient runs as two all-taps launches over
    # the row-parity views of x (vt_wgrad_span_s2_dispatch)
    (2, 32, 64, 3, 2, 56, 56),
    (2, 64, 128, 3, 2, 64, 72),
    (1, 16, 24, 3, 2, 60, 58),
    (3, 8, 16, 3, 2, 56, 64),
]


def _pad(k, s):
    return -((s - k) // 2)


STEM_CASES = [(2, 8, 32, 3, 1, 20, 13), (3, 8, 24, 3, 1, 17, 17), (2, 8, 32, 3, 1, 224, 224), (5, 8, 32, 3, 1, 31, 67),
              (1, 8, 16, 3, 1, 300, 500), (3, 8, 32, 3, 1, 2, 700)]


@pytest.mark.parametrize("tiles", [4, 2])
@pytest.mark.parametrize("case", STEM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_stem_tile_kernels_agree_with_the_one_tile_kernel(case, tiles):
    """vt_stem.hip: T tiles per workgroup + transposed product (round 3) against the round-1 kernel (VT_STEM_TILES=0), both
    passes: raw conv + statistics, and affine + ReLU.  Same MFMA dot products -> y bit-equal; the statistics are float
    sums of the same bf16 values in another order."""
    B, Cin, Cout, k, s, H, W = case
    x = filler.tensor(f"x{case}", (B, Cin, H, W))
    w = filler.tensor(f"w{case}", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    xd, wd = nhwc(x, N.VT_BF16), krsc(w, N.VT_BF16)
    sc = filler.tensor("stem_sc", (Cout,)).cuda().float().abs() + 0.5
    sf = filler.tensor("stem_sf", (Cout,)).cuda().float()
    ldy = Cout + 8  # a channel slice of a wider buffer
    out = {}
    try:
        for t in (0, tiles):
            N.set_knob("VT_STEM_TILES", t)
            yb = torch.full((B, H, W, ldy), 7.0, device="cuda", dtype=torch.bfloat16)
            ya = torch.full((B, H, W, ldy), 7.0, device="cuda", dtype=torch.bfloat16)
            st = N.stats_buffer(Cout)
            d = conv_desc(N.VT_BF16, xd, Cin, Cout, k, s, 1, ldy, flags=N.VT_CONV_STATS)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(yb), None, None, None, vp(st), stream()))
            name_s = N.last_kernel_name()
            d = conv_desc(N.VT_BF16, xd, Cin, Cout, k, s, 1, ldy, flags=N.VT_CONV_AFFINE | N.VT_CONV_RELU)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(ya), vp(sc), vp(sf), None, None, stream()))
            torch.cuda.synchronize()
            out[t] = (yb, ya, N.stats_decode(st), name_s)
    finally:
        N.set_knob("VT_STEM_TILES", 4)
    assert "stem_kernel" in out[0][3] and f"stem_t_kernel<{tiles}" in out[tiles][3]
    assert torch.equal(out[0][0], out[tiles][0]) and torch.equal(out[0][1], out[tiles][1])
    assert (out[0][0][..., Cout:] == 7.0).all() and (out[tiles][1][..., Cout:] == 7.0).all()  # nothing outside the slice
    a, b = out[0][2], out[tiles][2]
    assert ((a - b).abs() / (a.abs() + 1.0)).max().item() < 1e-5


@pytest.mark.parametrize("case", [(2, 8, 32, 3, 1, 20, 13), (3, 8, 24, 3, 1, 17, 17), (2, 8, 32, 3, 1, 224, 224)],
                         ids=lambda c: "x".join(map(str, c)))
def test_stem_statistics_only_pass(case):
    """VT_CONV_STATS | VT_CONV_NOSTORE (RGB stem kernels): the sums of the f32 accumulator (nothing is rounded because
    nothing is stored), against float64 on the bf16 operands; y (NULL here) is never touched; anywhere else the flag is
    refused."""
    B, Cin, Cout, k, s, H, W = case
    x = filler.tensor(f"x{case}", (B, Cin, H, W))
    w = filler.tensor(f"w{case}", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    ref = F.conv2d(rounded(x, N.VT_BF16).double(), rounded(w, N.VT_BF16).double(), None, s, 1)
    r1, r2 = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
    xd, wd = nhwc(x, N.VT_BF16), krsc(w, N.VT_BF16)
    try:
        for tiles in (4, 0):
            N.set_knob("VT_STEM_TILES", tiles)
            st = N.stats_buffer(Cout)
            d = conv_desc(N.VT_BF16, xd, Cin, Cout, k, s, 1, Cout, flags=N.VT_CONV_STATS | N.VT_CONV_NOSTORE)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), None, None, None, None, vp(st), stream()))
            got = N.stats_decode(st).cpu()
            np.testing.assert_allclose(got[0], r1, rtol=1e-5, atol=1e-5 * float(r2.max()) ** 0.5 * (B * H * W) ** 0.5)
            np.testing.assert_allclose(got[1], r2, rtol=1e-5)
    finally:
        N.set_knob("VT_STEM_TILES", 4)
    # not the stem shape class: refused, not silently stored
    x2 = nhwc(filler.tensor("xns", (1, 16, 8, 8)), N.VT_BF16)
    w2 = krsc(filler.tensor("wns", (16, 16, 3, 3)), N.VT_BF16)
    d = conv_desc(N.VT_BF16, x2, 16, 16, 3, 1, 1, 16, flags=N.VT_CONV_STATS | N.VT_CONV_NOSTORE)
    assert N.lib().vt_conv_igemm(C.byref(d), vp(x2), vp(w2), None, None, None, None, vp(N.stats_buffer(16)),
                                 stream()) == N.VT_ERR_UNSUPPORTED


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_forward_and_stats(dtype, case):
    B, Cin, Cout, k, s, H, W = case
    pad = _pad(k, s)
    x = filler.tensor(f"x{case}", (B, Cin, H, W))
    w = filler.tensor(f"w{case}", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, pad)
    xd, wd = nhwc(x, dtype), krsc(w, dtype)
    Ho, Wo = ref.shape[2:]
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda", dtype=TD[dtype])
    stats = N.stats_buffer(Cout)
    d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, Cout, flags=N.VT_CONV_STATS)
    N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), None, None, None, vp(stats), stream()))
    got = to_nchw(y)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < tol(dtype)
    # statistics are those of the values actually stored
    st = N.stats_decode(stats).cpu()
    yy = y.double().reshape(-1, Cout).cpu()
    np.testing.assert_allclose(st[0], yy.sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (yy * yy).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
def test_conv_channel_slices_and_fused_epilogue(dtype):
    """input and output are channel slices of wider (concat) buffers; eval epilogue
    y = relu(z*scale + shift) + residual in the same launch."""
    B, Cin, Cout, k, s, H, W = 2, 24, 40, 3, 1, 11, 6
    x = filler.tensor("sx", (B, Cin, H, W))
    w = filler.tensor("sw", (Cout, Cin, k, k), scale=0.1)
    r = filler.tensor("sr", (B, Cout, H, W))
    scale, shift = filler.tensor("ssc", (Cout,)) * 0.5 + 1.0, filler.tensor("ssh", (Cout,)) * 0.2
    z = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, 1)
    ref = torch.relu(z * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    if dtype == N.VT_BF16:
        ref = ref.to(torch.bfloat16).double()  # the fused path rounds before the residual add
    ref = ref + rounded(r, dtype).double()
    xd = nhwc(x, dtype, ld=64, coff=16)
    rd = nhwc(r, dtype, ld=48, coff=8)
    wide = torch.full((B, H, W, 96), float("nan"), device="cuda", dtype=TD[dtype])
    y = wide[..., 32 : 32 + Cout]
    wd = krsc(w, dtype)
    scd, shd = scale.cuda(), shift.cuda()  # keep alive: the launch is asynchronous
    flags = N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL
    d = conv_desc(dtype, xd, Cin, Cout, k, s, 1, 96, flags=flags, ldr=48)
    N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), vp(scd), vp(shd), vp(rd), None, stream()))
    assert rel_err(to_nchw(y), ref) < tol(dtype)
    # nothing outside the slice was written
    assert torch.isnan(wide[..., :32]).all() and torch.isnan(wide[..., 32 + Cout :]).all()


def test_conv_rejects_bad_arguments():
    x = torch.zeros(1, 4, 4, 8, device="cuda")
    d = conv_desc(N.VT_F32, x, 8, 6, 3, 1, 1, 6)  # Cout not a multiple of 4
    rc = N.lib().vt_conv_igemm(C.byref(d), vp(x), vp(x), vp(x), None, None, None, None, stream())
    assert rc == N.VT_ERR_UNSUPPORTED and "multiples" in N.last_error()
    d = conv_desc(N.VT_F32, x, 8, 8, 3, 1, 1, 8)
    rc = N.lib().vt_conv_igemm(C.byref(d), None, vp(x), vp(x), None, None, None, None, stream())
    assert rc == N.VT_ERR_INVALID
    # a successful launch so the autouse launch-count check holds
    N.check(N.lib().vt_memset(vp(x), 0, 16, stream()))


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_filter_gradient(dtype, case):
    B, Cin, Cout, k, s, H, W = case
    pad = _pad(k, s)
    x = rounded(filler.tensor(f"gx{case}", (B, Cin, H, W)), dtype).double()
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    z = F.conv2d(x, w, None, s, pad)
    dz = rounded(filler.tensor(f"gdz{case}", z.shape), dtype).double()
    z.backward(dz)
    xd, dzd = nhwc(x.float(), dtype), nhwc(dz.float(), dtype)
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    dw += 1.0  # the kernel accumulates into existing gradient
    d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, Cout)
    N.check(N.lib().vt_conv_wgrad(C.byref(d), vp(xd), vp(dzd), vp(dw), k * k * Cin, stream()))
    got = (dw - 1.0).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, w.grad) < (2e-5 if dtype == N.VT_F32 else 2e-5)  # inputs pre-rounded, f32 accumulate


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
@pytest.mark.parametrize("shape", [(2, 24, 5, 7), (3, 160, 4, 4), (2, 8, 16, 16), (6, 256, 7, 7), (5, 512, 5, 3), (3, 320, 9, 9)],
                         ids=str)
def test_batchnorm_relu_forward_backward_chain(dtype, shape):
    """stats -> finalize -> apply -> bwd reduce -> bwd finalize -> bwd apply vs autograd of
    F.batch_norm + relu (+ residual), incl. running statistics (components.py:36-44).  (256 / 512 / 320 channels on small
    maps: the backward reduction splits the channels over blockIdx.y there, round 4.)"""
    B, Cc, H, W = shape
    M = B * H * W
    z0 = rounded(filler.tensor(f"z{shape}", shape) * 1.5 + 0.3, dtype)
    res = rounded(filler.tensor(f"r{shape}", shape), dtype)
    gamma = (filler.tensor(f"g{shape}", (Cc,)) * 0.2 + 1.0).requires_grad_(True)
    beta = (filler.tensor(f"b{shape}", (Cc,)) * 0.2).requires_grad_(True)
    rm, rv = filler.tensor(f"rm{shape}", (Cc,)) * 0.1, filler.tensor(f"rv{shape}", (Cc,)).abs() + 0.5
    zt = z0.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = torch.relu(F.batch_norm(zt, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)) + res
    dy = rounded(filler.tensor(f"dy{shape}", shape), dtype)
    y_ref.backward(dy)

    zd, rd, dyd = nhwc(z0, dtype), nhwc(res, dtype), nhwc(dy, dtype)
    stats = N.stats_buffer(Cc)
    zz = zd.float().reshape(-1, Cc)
    N.stats_encode(stats, 0, zz.sum(0), replica=3)  # (any replica: the finalize kernel sums them all)
    N.stats_encode(stats, 1, (zz * zz).sum(0), replica=7)
    g, b_ = gamma.detach().cuda(), beta.detach().cuda()
    rmd, rvd = rm.cuda(), rv.cuda()
    nbt = torch.zeros(2, dtype=torch.int64, device="cuda")
    coef = torch.zeros(4, Cc, device="cuda")
    L = N.lib()
    N.check(L.vt_bn_finalize(vp(stats), Cc, float(M), vp(g), vp(b_), 1e-5, 0.1, vp(rmd), vp(rvd), vp(nbt),
                             vp(coef[0]), vp(coef[1]), vp(coef[2]), vp(coef[3]), stream()))
    y = torch.empty_like(zd)
    N.check(L.vt_bn_act_apply(vp(zd), Cc, vp(coef[0]), vp(coef[1]), vp(rd), Cc, vp(y), Cc, M, Cc, 1, dtype, stream()))
    assert rel_err(to_nchw(y), y_ref.detach()) < tol(dtype, 1e-5)
    np.testing.assert_allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu(), rv_ref, rtol=1e-4, atol=1e-6)
    assert nbt[0].item() == 1
    sums = N.stats_buffer(Cc)
    N.check(L.vt_bn_act_bwd_reduce(vp(dyd), Cc, vp(zd), Cc, vp(coef[0]), vp(coef[1]), vp(coef[2]), vp(coef[3]),
                                   M, Cc, 1, dtype, vp(sums), stream()))
    dg, db = torch.ones(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    bc = torch.zeros(3, Cc, device="cuda")
    N.check(L.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 1.0, vp(coef[0]), vp(coef[2]), vp(coef[3]), 1, vp(dg), vp(db),
                                 vp(bc), stream()))
    dz = torch.empty_like(zd)
    N.check(L.vt_bn_act_bwd_apply(vp(dyd), Cc, vp(zd), Cc, vp(coef[0]), vp(coef[1]), vp(bc), vp(dz), Cc, M, Cc, 1,
                                  dtype, stream()))
    assert rel_err(to_nchw(dz), zt.grad) < tol(dtype, 1e-4)
    np.testing.assert_allclose((dg - 1).cpu(), gamma.grad, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose((db - 1).cpu(), beta.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
@pytest.mark.parametrize("shape", [(2, 16, 9, 9), (1, 40, 8, 12), (2, 8, 7, 7)], ids=str)
def test_maxpool_3x3_s2(dtype, shape):
    """nn.MaxPool2d(3,2,1) (vovnet.py:94), ties included: the input is post-ReLU."""
    x = rounded(torch.relu(filler.tensor(f"mp{shape}", shape)), dtype).requires_grad_(True)
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = rounded(filler.tensor(f"mpdy{shape}", y_ref.shape), dtype)
    y_ref.backward(dy)
    B, Cc, H, W = shape
    xd, dyd = nhwc(x.detach(), dtype), nhwc(dy, dtype)
    Ho, Wo = y_ref.shape[2:]
    y = torch.empty(B, Ho, Wo, Cc, device="cuda", dtype=TD[dtype])
    am = torch.zeros(B * Ho * Wo * Cc, dtype=torch.uint8, device="cuda")
    L = N.lib()
    N.check(L.vt_maxpool3x3s2_fwd(vp(xd), Cc, vp(y), Cc, vp(am), B, H, W, Cc, dtype, stream()))
    assert torch.equal(to_nchw(y), y_ref.detach())
    dx = torch.full((B, H, W, Cc), 1.0, device="cuda", dtype=TD[dtype])
    N.check(L.vt_maxpool3x3s2_bwd(vp(dyd), Cc, vp(am), vp(dx), Cc, B, H, W, Cc, 1, dtype, stream()))
    assert rel_err(to_nchw(dx) - 1.0, x.grad) < tol(dtype, 1e-6, 1e-2)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
def test_global_avgpool_and_ese_gate(dtype):
    B, Cc, H, W = 3, 24, 5, 4
    x = rounded(filler.tensor("ex", (B, Cc, H, W)), dtype).requires_grad_(True)
    s = rounded(filler.tensor("es", (B, Cc, 1, 1)) * 3, dtype).requires_grad_(True)
    r = rounded(filler.tensor("er", (B, Cc, H, W)), dtype)
    y_ref = x * F.hardsigmoid(s) + r
    dy = rounded(filler.tensor("edy", y_ref.shape), dtype)
    y_ref.backward(dy)
    L = N.lib()
    xd, sd, rd, dyd = nhwc(x.detach(), dtype), nhwc(s.detach(), dtype), nhwc(r, dtype), nhwc(dy, dtype)
    pooled = torch.empty(B, Cc, device="cuda", dtype=TD[dtype])
    N.check(L.vt_global_avgpool_fwd(vp(xd), Cc, vp(pooled), Cc, B, H * W, Cc, dtype, stream()))
    assert rel_err(pooled.float().cpu(), x.detach().mean((2, 3))) < tol(dtype, 1e-6)
    y = torch.empty_like(xd)
    N.check(L.vt_ese_gate_fwd(vp(xd), Cc, vp(sd), Cc, vp(rd), Cc, vp(y), Cc, B, H * W, Cc, dtype, stream()))
    assert rel_err(to_nchw(y), y_ref.detach()) < tol(dtype, 1e-6)
    dx = torch.zeros_like(xd)
    ds = torch.zeros(B, Cc, device="cuda")
    N.check(L.vt_ese_gate_bwd(vp(dyd), Cc, vp(xd), Cc, vp(sd), Cc, vp(dx), Cc, vp(ds), B, H * W, Cc, 0, dtype, stream()))
    assert rel_err(to_nchw(dx), x.grad) < tol(dtype, 1e-6)
    assert rel_err(ds.cpu(), s.grad.reshape(B, Cc)) < 1e-5
    # avgpool backward with accumulation
    g = rounded(filler.tensor("apg", (B, Cc)), dtype)
    gd = g.to("cuda", TD[dtype])
    acc = torch.ones(B, H, W, Cc, device="cuda", dtype=TD[dtype])
    N.check(L.vt_global_avgpool_bwd(vp(gd), Cc, vp(acc), Cc, B, H * W, Cc, 1, dtype, stream()))
    ref = 1.0 + (g / (H * W))[:, :, None, None].expand(B, Cc, H, W)
    assert rel_err(to_nchw(acc), ref) < tol(dtype, 1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
def test_softmax_cross_entropy_label_smoothing(dtype):
    """F.cross_entropy(logits, labels, label_smoothing) (classifier.py:92) and its gradient."""
    B, Ncls = 6, 1000
    logits = rounded(filler.tensor("xl", (B, Ncls)) * 3, dtype).requires_grad_(True)
    y = filler.labels(B, Ncls)
    loss_ref = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss_ref.backward()
    ld = logits.detach().to("cuda", TD[dtype])
    loss = torch.zeros(1, device="cuda")
    dl = torch.empty_like(ld)
    yd = y.cuda()
    N.check(N.lib().vt_softmax_xent(vp(ld), Ncls, vp(yd), 0.1, 1.0 / B, vp(loss), vp(dl), Ncls, B, Ncls,
                                    dtype, stream()))
    assert loss.item() == pytest.approx(loss_ref.item(), rel=1e-5)
    assert rel_err(dl.float().cpu(), logits.grad) < tol(dtype, 1e-5)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
def test_validation_cross_entropy_and_top1(dtype):
    """vt_softmax_xent_eval: loss sum without label smoothing, top-1 hits (first maximum, as torch.argmax) and rows,
    accumulated over two calls (classifier.py:97-109)."""
    B, Ncls = 37, 1000
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(B, Ncls, generator=g) * 3).to(TD[dtype])
    logits[3, 10] = logits[3, 700] = 40.0  # a tie: the first maximum wins
    labels = torch.randint(0, Ncls, (B,), generator=g)
    labels[3] = 10
    labels[4] = int(logits[4].float().argmax())
    ld = Ncls + 8
    dev = torch.zeros(B, ld, dtype=TD[dtype], device="cuda")
    dev[:, :Ncls] = logits.cuda()
    out = torch.zeros(3, device="cuda")
    lab = labels.cuda()
    for _ in range(2):
        N.check(N.lib().vt_softmax_xent_eval(vp(dev), ld, vp(lab), vp(out), B, Ncls, dtype, stream()))
    torch.cuda.synchronize()
    ref = F.cross_entropy(logits.double(), labels, reduction="sum").item()
    hits = int((logits.float().argmax(-1) == labels).sum())
    assert hits >= 2
    assert abs(out[0].item() - 2 * ref) < 1e-4 * abs(2 * ref)
    assert out[1].item() == 2 * hits and out[2].item() == 2 * B
    bad = lab.clone()
    bad[0] = Ncls  # out of range: torch raises, the kernel poisons the sum
    N.check(N.lib().vt_softmax_xent_eval(vp(dev), ld, vp(bad), vp(out), B, Ncls, dtype, stream()))
    assert torch.isnan(out[0]).item()


def test_softmax_cross_entropy_out_of_range_label_poisons_the_loss():
    """torch raises on a label outside [0, N); the kernel cannot, so it must neither read out of bounds
    nor train silently: the loss becomes NaN, the gradients of the valid samples stay finite."""
    B, Ncls = 4, 10
    ld = filler.tensor("xo", (B, Ncls)).cuda()
    loss = torch.zeros(1, device="cuda")
    dl = torch.empty_like(ld)
    yd = torch.tensor([1, 12345678, 3, -5], dtype=torch.int64, device="cuda")
    N.check(N.lib().vt_softmax_xent(vp(ld), Ncls, vp(yd), 0.1, 1.0 / B, vp(loss), vp(dl), Ncls, B, Ncls,
                                    N.VT_F32, stream()))
    assert torch.isnan(loss).all()
    assert torch.isfinite(dl).all()


def test_sgd_momentum_matches_torch_optim():
    n = 100_003
    p0, g = filler.tensor("sp", (n,)), filler.tensor("sg", (n,))
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.05, momentum=0.9, weight_decay=2e-5)
    n_pad = (n + 3) // 4 * 4
    p = torch.zeros(n_pad, device="cuda")
    p[:n] = p0.cuda()
    gd = torch.zeros(n_pad, device="cuda")
    gd[:n] = g.cuda()
    m = torch.zeros(n_pad, device="cuda")
    mirror = torch.zeros(n_pad, device="cuda", dtype=torch.bfloat16)
    lr_dev = torch.tensor([0.05], device="cuda")
    for step in range(3):
        ref.grad = g.clone() * (step + 1)
        opt.step()
        N.check(N.lib().vt_sgd_momentum(vp(p), vp(gd), vp(m), vp(mirror), N.VT_BF16, n, 123.0, 0.9, 2e-5,
                                        float(step + 1), vp(lr_dev), stream()))
    torch.testing.assert_close(p[:n].cpu(), ref.data, rtol=1e-6, atol=1e-6)  # fma vs mul+add
    assert torch.equal(mirror[:n], p[:n].to(torch.bfloat16))  # the mirror is the RNE cast of the new weights


def test_layout_and_pack_kernels():
    L = N.lib()
    x = filler.tensor("img", (2, 3, 5, 7))
    for dtype in DTYPES:
        cpad = 8 if dtype == N.VT_BF16 else 4
        y = torch.full((2, 5, 7, cpad), float("nan"), device="cuda", dtype=TD[dtype])
        xc = x.cuda()
        N.check(L.vt_nchw_to_nhwc(vp(xc), vp(y), 2, 3, 5, 7, cpad, dtype, stream()))
        assert torch.equal(y[..., :3].float().cpu(), rounded(x, dtype).permute(0, 2, 3, 1))
        assert (y[..., 3:] == 0).all()
        back = torch.zeros(2, 3, 5, 7, device="cuda")
        N.check(L.vt_nhwc_to_nchw(vp(y), cpad, vp(back), 2, 3, 5, 7, dtype, stream()))
        assert torch.equal(back.cpu(), rounded(x, dtype))
    # copy2d with conversion and accumulation
    a = filler.tensor("c2", (6, 10))
    dst = torch.ones(6, 16, device="cuda")
    ac = a.cuda()
    N.check(L.vt_copy2d(vp(ac), N.VT_F32, 10, vp(dst), N.VT_F32, 16, 6, 10, 1, stream()))
    torch.testing.assert_close(dst[:, :10].cpu(), a + 1)
    assert (dst[:, 10:] == 1).all()
    # dgrad filter pack: out[c][i][n] = w[n][sel[i]][c]
    Cout, taps, Cin = 8, 9, 4
    w = filler.tensor("pk", (Cout, taps, Cin))
    sel = [8, 6, 2, 0]
    out = torch.zeros(Cin, len(sel), Cout, device="cuda", dtype=torch.bfloat16)
    arr = (C.c_int32 * len(sel))(*sel)
    wc = w.cuda()
    N.check(L.vt_pack_dgrad_filter(vp(wc), N.VT_F32, taps * Cin, vp(out), N.VT_BF16, arr, len(sel), Cout, taps,
                                   Cin, stream()))
    ref = w[:, sel, :].permute(2, 1, 0).to(torch.bfloat16)
    assert torch.equal(out.cpu(), ref)
    # column sums
    mat = filler.tensor("cs", (37, 24))
    acc = torch.ones(24, device="cuda")
    mc = mat.cuda()
    N.check(L.vt_colsum(vp(mc), 24, 37, 24, N.VT_F32, vp(acc), stream()))
    torch.testing.assert_close(acc.cpu(), mat.sum(0) + 1, rtol=1e-5, atol=1e-5)


def test_executor_and_graph_replay_agree():
    """vt_run_ops and a captured hipGraph of the same list give identical results."""
    from vision_toolbox import engine as E

    n = 4096
    src = filler.tensor("gsrc", (n,)).cuda()
    dst1, dst2 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    op = N.Op()
    op.kind = N.OP_COPY2D
    for k in range(N.VT_OP_MAX_PTR):
        op.ptr[k].base = -1
    op.ptr[0].base, op.ptr[1].base = 0, 1
    op.i[0], op.i[1], op.i[2], op.i[3] = N.VT_F32, N.VT_F32, n, 1
    op.f[0], op.f[1], op.f[2] = n, n, 1
    ops = E.ops_array([op, op])
    N.run_ops(ops, 2, [src.data_ptr(), dst1.data_ptr()], stream())
    g = N.Graph(ops, 2, [src.data_ptr(), dst2.data_ptr()])
    g.launch(stream())
    torch.cuda.synchronize()
    assert torch.equal(dst1, 2 * src) and torch.equal(dst2, 2 * src)
    g.launch(stream())
    torch.cuda.synchronize()
    assert torch.equal(dst2, 4 * src)
    # error surface: unknown kind is reported, not fatal
    bad = N.Op()
    bad.kind = 999
    with pytest.raises(N.NativeError):
        N.run_ops(E.ops_array([bad]), 1, [src.data_ptr()], stream())


def test_fork_behind_a_kernel_completion_event_orders_the_side_stream():
    """The executor releases the side stream at a FORK with the completion event of the kernel right before it
    (hipExtLaunchKernelGGL stop event, vt_runtime.hip) instead of a recorded event.  A long BatchNorm apply on the
    main stream, then FORK, then a short column sum of ITS OUTPUT on the side stream: were the side stream released
    early, the sum would see the previous round's output (the rounds differ by their scale)."""
    from vision_toolbox import engine as E

    M, Cc = 1 << 20, 64
    z = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
    y = torch.zeros_like(z)
    scale, shift = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    out = torch.zeros(Cc, device="cuda")

    def blank(kind, side=False):
        op = N.Op()
        op.kind = kind | (N.OP_SIDE_STREAM if side else 0)
        for k in range(N.VT_OP_MAX_PTR):
            op.ptr[k].base = -1
        return op

    a = blank(N.OP_BN_ACT_APPLY)  # ptr: z scale shift residual y | i: ldz ldr ldy C relu dtype | f: M
    a.ptr[0].base, a.ptr[1].base, a.ptr[2].base, a.ptr[4].base = 0, 1, 2, 3
    a.i[0], a.i[1], a.i[2], a.i[3], a.i[4], a.i[5] = Cc, 0, Cc, Cc, 0, N.VT_BF16
    a.f[0] = M
    c = blank(N.OP_COLSUM, side=True)  # ptr: a out | i: lda C dtype fixed | f: M
    c.ptr[0].base, c.ptr[1].base = 3, 4
    c.i[0], c.i[1], c.i[2], c.i[3] = Cc, Cc, N.VT_BF16, 0
    c.f[0] = M
    ops = E.ops_array([a, blank(N.OP_FORK), c, blank(N.OP_JOIN)])
    bases = [z.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), out.data_ptr()]
    side = torch.cuda.Stream()
    for r in range(1, 7):
        scale.fill_(float(r))  # exact in bf16: y = r z
        out.zero_()
        torch.cuda.synchronize()
        N.run_ops(ops, 4, bases, stream(), side=int(side.cuda_stream))
        torch.cuda.synchronize()
        want = (z.float() * r).to(torch.bfloat16).float().sum(0)
        torch.testing.assert_close(out, want, rtol=1e-4, atol=0.05 * r)


@pytest.mark.parametrize("shape", [(2, 24, 20), (1, 8, 130), (3, 10, 256), (1, 64, 320), (2, 6, 2)], ids=str)
@pytest.mark.parametrize("flags", [0, N.VT_CONV_AFFINE | N.VT_CONV_RELU, N.VT_CONV_AFFINE], ids=["raw", "affine_relu", "affine"])
def test_yolov5_stem_kernel(shape, flags):
    """6x6 stride-2 conv over padded-RGB pixels to 80 channels (the Darknet-YOLOv5x stem, vt_stem6.hip): against the
    float64 convolution and against the gather kernel it replaces (same K order: the outputs must agree bit for bit),
    on maps that end inside a 4 x 64 output block, with the output as a channel slice of a wider tensor."""
    B, H, W = shape
    dtype, Cin, Cout, k, s, pad = N.VT_BF16, 8, 80, 6, 2, 2
    x = filler.tensor(f"s6x{shape}", (B, Cin, H, W))
    x[:, 3:] = 0
    w = filler.tensor(f"s6w{shape}", (Cout, Cin, k, k), scale=(2.0 / (3 * k * k)) ** 0.5)
    sc = filler.tensor("s6s", (Cout,)).abs() + 0.5
    sf = filler.tensor("s6f", (Cout,)) * 0.2
    ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, pad)
    if flags & N.VT_CONV_AFFINE:
        ref = ref * sc.double()[None, :, None, None] + sf.double()[None, :, None, None]
    if flags & N.VT_CONV_RELU:
        ref = torch.relu(ref)
    xd, wd, scd, sfd = nhwc(x, dtype), krsc(w, dtype), sc.cuda(), sf.cuda()
    Ho, Wo = ref.shape[2:]
    ldy = 96
    outs, names = [], []
    try:
        for on in (1, 0):
            N.set_knob("VT_STEM6_KERNEL", on)
            y = torch.full((B, Ho, Wo, ldy), 7.0, device="cuda", dtype=torch.bfloat16)
            d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, ldy, flags=flags)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), vp(scd) if flags else None, vp(sfd) if flags else None,
                                          None, None, stream()))
            torch.cuda.synchronize()
            names.append(N.last_kernel_name())
            outs.append(y)
    finally:
        N.set_knob("VT_STEM6_KERNEL", 1)
    assert "stem6_kernel" in names[0] and "igemm_kernel" in names[1], names
    assert (outs[0][..., Cout:] == 7.0).all()  # nothing outside the channel slice
    assert rel_err(to_nchw(outs[0][..., :Cout]), ref) < tol(dtype)
    assert torch.equal(outs[0], outs[1])


RAGGED_K_CASES = [(2, 80, 80, 3, 1, 12, 12), (2, 80, 160, 3, 2, 16, 16), (3, 80, 80, 1, 1, 10, 7), (1, 48, 80, 3, 1, 9, 9),
                  (1, 80, 160, 3, 1, 20, 20), (2, 160, 80, 1, 1, 13, 9), (2, 8, 80, 6, 2, 24, 20)]


@pytest.mark.parametrize("case", RAGGED_K_CASES, ids=lambda c: "x".join(map(str, c)))
def test_gather_kernel_80_wide_tiles(case):
    """80- and 160-channel outputs (Darknet-YOLOv5x) take 80- / 160-wide filter tiles on the gather kernel (VT_IGEMM_BN80;
    VT_IGEMM_W8: 8-wave workgroups on 256 x 160 and 256 x 128 tiles): every tile shape against the float64 convolution,
    training and fused inference epilogue (with residual), VT_IGEMM_SPAN=0 / VT_SPAN6=0 keeping the input-span
    kernels out."""
    dtype = N.VT_BF16
    B, Cin, Cout, k, s, H, W = case
    pad = _pad(k, s)
    x = filler.tensor(f"rx{case}", (B, Cin, H, W))
    w = filler.tensor(f"rw{case}", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    sc = filler.tensor(f"rs{case}", (Cout,)).abs() + 0.5
    sf = filler.tensor(f"rf{case}", (Cout,)) * 0.1
    ref = F.conv2d(rounded(x, dtype).double(), rounded(w, dtype).double(), None, s, pad)
    Ho, Wo = ref.shape[2:]
    res = filler.tensor(f"rr{case}", (B, Cout, Ho, Wo))
    ref_aff = torch.relu(ref * sc.double()[None, :, None, None] + sf.double()[None, :, None, None]) + rounded(res, dtype).double()
    xd, wd, rd = nhwc(x, dtype), krsc(w, dtype), nhwc(res, dtype)
    scd, sfd = sc.cuda(), sf.cuda()
    names, outs = [], []
    try:
        N.set_knob("VT_IGEMM_SPAN", 0)
        N.set_knob("VT_SPAN6", 0)
        for bn80, w8 in ((0, 0), (1, 0), (1, 7), (0, 6)):  # (bit 2: the 8-wave tile on small maps too)
            N.set_knob("VT_IGEMM_BN80", bn80)
            N.set_knob("VT_IGEMM_W8", w8)
            y = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda", dtype=TD[dtype])
            stats = N.stats_buffer(Cout)
            d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, Cout, flags=N.VT_CONV_STATS)
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(y), None, None, None, vp(stats), stream()))
            names.append(N.last_kernel_name())
            assert rel_err(to_nchw(y), ref) < tol(dtype), names
            st = N.stats_decode(stats).cpu()
            yy = y.double().reshape(-1, Cout).cpu()
            np.testing.assert_allclose(st[0], yy.sum(0), rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(st[1], (yy * yy).sum(0), rtol=1e-4, atol=1e-3)
            ya = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda", dtype=TD[dtype])
            d = conv_desc(dtype, xd, Cin, Cout, k, s, pad, Cout, flags=N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL)
            d.ldr = Cout
            N.check(N.lib().vt_conv_igemm(C.byref(d), vp(xd), vp(wd), vp(ya), vp(scd), vp(sfd), vp(rd), None, stream()))
            assert rel_err(to_nchw(ya), ref_aff) < tol(dtype), names
            outs.append((y, ya))
    finally:
        N.set_knob("VT_IGEMM_SPAN", 1)
        N.set_knob("VT_SPAN6", 1)
        N.set_knob("VT_IGEMM_BN80", 1)
        N.set_knob("VT_IGEMM_W8", 3)
    assert ",256,80," in names[1] and ",80," not in names[0] and ",160," not in names[0], names
    assert (",256,160,4,2," if Cout == 160 else ",256,80,") in names[2] and ",256,128,4,2," in names[3], names
    # same K order, same accumulators: the tile shape does not change a single output
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


@pytest.mark.parametrize("mode", ["0", "2", "3"], ids=["general_only", "span_forced", "span_256row_tiles"])
def test_conv_kernel_variants_in_subprocess(mode):
    """vt_conv_igemm picks between the general gather kernel and the input-span kernel (and its
    tile heights) from the problem size; the sizes above only reach some of them.  The choice
    is read from VT_IGEMM_SPAN once per process, so the conv tests are re-run in a child
    process per setting: 0 = general kernel everywhere, 2 = span kernel wherever it applies,
    3 = span kernel with 256-row tiles."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, VT_IGEMM_SPAN=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "test_conv_forward_and_stats or test_conv_channel_slices_and_fused_epilogue"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # the autouse fixture wants a launch from THIS process too
    buf = torch.zeros(16, device="cuda")
    N.check(N.lib().vt_memset(vp(buf), 0, 64, stream()))


def test_stride2_all_taps_filter_gradient_in_subprocess():
    """the stride-2 filter gradient as two all-taps launches over the row-parity views of x (vt_wgrad_span_s2_dispatch)
    is taken from 100 output columns and 32 input channels up (the first stride-2 conv of the Darknets at 224 px:
    test_fullsize_gpu.py covers that shape at batch 256).  VT_WGRAD_S2_MINW / _MINC are read once per process, so the
    filter-gradient cases above (28 .. 36 output columns, 8 .. 64 channels, one and two channel tiles) are re-run in a
    child process with the thresholds lowered."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, VT_WGRAD_S2_MINW="28", VT_WGRAD_S2_MINC="8")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k", "test_conv_filter_gradient"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    buf = torch.zeros(16, device="cuda")
    N.check(N.lib().vt_memset(vp(buf), 0, 64, stream()))


def test_filter_repack_batch_equals_the_single_launches():
    """vt_pack_dgrad_filter_batch (what the executor turns a run of VT_OP_PACK_DGRAD ops into): 45 filters of mixed shapes
    -- more than one VT_PACK_BATCH, 1x1 / 3x3 / depth-to-space tables with zero taps, a channel-slice source (ldw > taps *
    Cin) -- bit-equal to one vt_pack_dgrad_filter call each."""
    L = N.lib()
    gen = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(64, 32, 9), (128, 128, 1), (32, 64, 9), (256, 128, 9), (80, 80, 9), (8, 24, 1), (128, 64, 9)]
    items = (N.PackItem * 45)()
    keep, ref = [], []
    for k in range(45):
        Cout, Cin, taps = shapes[k % len(shapes)]
        ldw = taps * Cin + (16 if k % 3 == 0 else 0)
        w = torch.randn(Cout, ldw, device="cuda", generator=gen).to(torch.bfloat16)
        if taps == 9 and k % 2:  # a depth-to-space table: 4 column blocks x 4 taps, -1 where a class has no tap
            sel = [4, -1, -1, -1, 5, 3, -1, -1, 7, -1, 1, -1, 8, 6, 2, 0]
        else:
            sel = list(range(taps - 1, -1, -1))
        out1 = torch.full((Cin * len(sel) * Cout,), 7.0, device="cuda", dtype=torch.bfloat16)
        out2 = torch.full_like(out1, 3.0)
        arr = (C.c_int32 * len(sel))(*sel)
        N.check(L.vt_pack_dgrad_filter(vp(w), N.VT_BF16, ldw, vp(out1), N.VT_BF16, arr, len(sel), Cout, taps, Cin, stream()))
        it = items[k]
        it.w, it.out, it.ldw, it.nsel, it.Cout, it.ntaps, it.Cin = w.data_ptr(), out2.data_ptr(), ldw, len(sel), Cout, taps, Cin
        for i, v in enumerate(sel):
            it.sel[i] = v
        keep.append((w, out2))
        ref.append(out1)
    N.check(L.vt_pack_dgrad_filter_batch(C.byref(items), 45, stream()))
    torch.cuda.synchronize()
    for (w, out2), out1 in zip(keep, ref):
        assert torch.equal(out1, out2)


@pytest.mark.parametrize("dtype", DTYPES, ids=DNAME.get)
@pytest.mark.parametrize("shape", [(2, 16, 12, 10), (3, 24, 7, 9), (1, 64, 56, 56), (2, 8, 2, 2), (2, 40, 15, 16)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("with_res", [False, True], ids=["plain", "residual"])
def test_bn_apply_fused_with_the_max_pool_and_its_backward(dtype, shape, with_res):
    """Round 4: vt_bn_act_apply_pool / vt_bn_act_bwd_reduce_pool / vt_bn_act_bwd_apply_pool (the normalise pass of the unit
    whose output VoVNet's `stage.max_pool` reads, vovnet.py:94, fused with that pool; the unit's BatchNorm backward reading
    the POOLED gradient through the arg-max taps) against the separate launches they replace -- vt_bn_act_apply +
    vt_maxpool3x3s2_fwd, vt_maxpool3x3s2_bwd + vt_bn_act_bwd_reduce / _apply -- on the same operands: every tensor
    output bit-equal (the same roundings at the same points), the channel sums to f32 summation order; channel-slice
    operands, odd maps, a map smaller than a window."""
    B, Cc, H, W = shape
    L = N.lib()
    epc = 8 if dtype == N.VT_BF16 else 4
    if Cc % epc:
        pytest.skip("channel count not a multiple of the 16-byte chunk")
    z = nhwc(filler.tensor(f"fpz{shape}", shape), dtype, ld=Cc + 2 * epc, coff=epc)
    res = nhwc(filler.tensor(f"fpr{shape}", shape), dtype) if with_res else None
    scale = (filler.tensor(f"fps{shape}", (Cc,)).abs() + 0.5).cuda()
    shift = (filler.tensor(f"fpf{shape}", (Cc,)) * 0.3).cuda()
    mean = (filler.tensor(f"fpm{shape}", (Cc,)) * 0.1).cuda()
    invstd = (filler.tensor(f"fpi{shape}", (Cc,)).abs() + 0.5).cuda()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ldz = z.stride(2)

    def fresh(h, w, c=Cc):
        return torch.full((B, h, w, c + epc), float("nan"), device="cuda", dtype=TD[dtype])

    # reference: separate launches
    y0, p0 = fresh(H, W), fresh(Ho, Wo)
    am0 = torch.zeros(B * Ho * Wo * Cc, dtype=torch.uint8, device="cuda")
    N.check(L.vt_bn_act_apply(vp(z), ldz, vp(scale), vp(shift), vp(res), Cc, vp(y0), Cc + epc, B * H * W, Cc, 1, dtype, stream()))
    N.check(L.vt_maxpool3x3s2_fwd(vp(y0), Cc + epc, vp(p0), Cc + epc, vp(am0), B, H, W, Cc, dtype, stream()))
    # fused
    y1, p1 = fresh(H, W), fresh(Ho, Wo)
    am1 = torch.zeros_like(am0)
    N.check(L.vt_bn_act_apply_pool(vp(z), ldz, vp(scale), vp(shift), vp(res), Cc, vp(y1), Cc + epc, vp(p1), Cc + epc, vp(am1), B, H, W,
                                   Cc, 1, dtype, stream()))
    torch.cuda.synchronize()
    for a, b in ((y0, y1), (p0, p1)):
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    assert torch.equal(am0, am1)
    # backward: the pooled gradient through the taps
    dp = nhwc(filler.tensor(f"fpd{shape}", (B, Cc, Ho, Wo)), dtype)
    dy = torch.zeros(B, H, W, Cc, device="cuda", dtype=TD[dtype])
    N.check(L.vt_maxpool3x3s2_bwd(vp(dp), Cc, vp(am0), vp(dy), Cc, B, H, W, Cc, 0, dtype, stream()))
    s0, s1 = N.stats_buffer(Cc), N.stats_buffer(Cc)
    M = B * H * W
    N.check(L.vt_bn_act_bwd_reduce(vp(dy), Cc, vp(z), ldz, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, 1, dtype, vp(s0), stream()))
    N.check(L.vt_bn_act_bwd_reduce_pool(vp(dp), Cc, vp(am1), vp(z), ldz, vp(scale), vp(shift), vp(mean), vp(invstd), B, H, W, Cc, 1,
                                        dtype, vp(s1), stream()))
    torch.cuda.synchronize()
    # (per-thread f32 partial sums run over different row sets in the two kernels: equal to f32 summation noise)
    d0, d1 = N.stats_decode(s0), N.stats_decode(s1)
    assert torch.allclose(d0, d1, rtol=2e-5, atol=2e-5 * float(d0.abs().max()))
    coef = (filler.tensor(f"fpc{shape}", (3 * Cc,)) * 0.5).cuda()
    dz0, dz1 = fresh(H, W), fresh(H, W)
    N.check(L.vt_bn_act_bwd_apply(vp(dy), Cc, vp(z), ldz, vp(scale), vp(shift), vp(coef), vp(dz0), Cc + epc, M, Cc, 1, dtype, stream()))
    N.check(L.vt_bn_act_bwd_apply_pool(vp(dp), Cc, vp(am1), vp(z), ldz, vp(scale), vp(shift), vp(coef), vp(dz1), Cc + epc, B, H, W, Cc,
                                       1, dtype, stream()))
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(dz0), torch.isnan(dz1)) and torch.equal(torch.nan_to_num(dz0.float()), torch.nan_to_num(dz1.float()))


def test_collective_entry_points_through_the_c_abi_in_subprocess():
    """vt_comm_unique_id / vt_comm_init / vt_allreduce_bucket / vt_stat_sync / vt_comm_destroy (include/vt_amd.h; DDP's
    bucket all-reduce and SyncBatchNorm's statistics exchange, configs/base.yaml:17-22) driven through ctypes alone over a
    one-rank communicator, their error returns included (tools/comm_abi_check.py; a child process: the communicator is
    per process)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "comm_abi_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COMM_ABI_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    buf = torch.zeros(16, device="cuda")
    N.check(N.lib().vt_memset(vp(buf), 0, 64, stream()))
