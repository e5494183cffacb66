"""Kernel-level parity of the pointwise (1x1) ConvNormAct kernels (vt_pointwise.hip) through the C-ABI.

The four passes (statistics, normalise, backward reduction, backward apply + data / filter gradient) recompute
z = W x instead of reading a stored z; the reference here is torch float64 on the SAME bf16 operands with z rounded
to bf16 where the unfused path stores it (components.py:26-44 and its autograd backward).  What differs from the
kernels is the f32 summation order inside z, which moves a few z values by one bf16 ulp: sums are compared to 1e-3 of
their scale, element-wise outputs to a relative L2 of 4e-3 (one bf16 rounding of the output) with a bound on the
worst element.  Covered: one and two output groups (CSP conv1 | conv2), channel-slice operands (ld > C), pixel counts
that are not multiples of the 16-pixel tile, relu on / off, with and without residual / addend, shapes whose filter
gradient is formed inside the kernel (N*K <= 4096) and shapes that hand dz to vt_conv_wgrad.
"""
import ctypes as C

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import rel_err, stream

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def _lib_loaded():
    N.lib()
    before = N.launch_count()
    yield
    torch.cuda.synchronize()
    assert N.launch_count() > before, "no libvt_amd launch happened: the HIP path did not run"


def _rows(M, Cc, slack, gen, scale=1.0):
    """[M][Cc] bf16 rows inside a wider NaN-filled buffer when slack > 0 (a channel slice of a concat buffer)"""
    wide = torch.full((M, Cc + slack), float("nan"), device="cuda", dtype=BF)
    view = wide[:, slack // 2: slack // 2 + Cc] if slack else wide
    view.copy_((torch.randn(M, Cc, device="cuda", generator=gen) * scale).to(BF))
    return view


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _vps(ts):
    return _arr(C.c_void_p, [C.c_void_p(t.data_ptr()) if t is not None else None for t in ts])


def _desc(x, ws, relu):
    d = N.PwDesc()
    d.dtype, d.K, d.ngroups, d.relu, d.M = N.VT_BF16, ws[0].shape[1], len(ws), int(relu), x.shape[0]
    d.x, d.ldx = x.data_ptr(), x.stride(0)
    for g, w in enumerate(ws):
        d.C[g], d.w[g], d.ldw[g] = w.shape[0], w.data_ptr(), w.stride(0)
    return d


CASES = [
    # (M, K, [C per group], slack, relu)
    (16 * 37, 32, [32], 0, True),
    (1000, 32, [32], 16, True),      # tail tile (1000 = 62 * 16 + 8), slices
    (4096, 64, [32, 32], 0, True),   # CSP stage 0: conv1 | conv2 on 64 channels
    (2085, 64, [64], 32, False),
    (3000, 64, [32], 0, True),       # Darknet-53 block conv1 (C -> C/2)
    (2048, 128, [64, 64], 16, True),  # CSP stage 1 (filter gradient outside the kernel)
    (1555, 128, [128], 0, True),
    (1024, 128, [64], 0, False),
    (1555, 160, [160], 0, True),     # Darknet-YOLOv5x bottleneck conv1 (inference runs the apply pass alone)
    (2048, 160, [160], 16, False),
    (1024, 160, [96, 64], 0, True),
    (1100, 320, [160], 0, True),     # Darknet-YOLOv5x CSP conv1 / conv2 of the 320-channel stage
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c[0]}_K{c[1]}_C{'+'.join(map(str, c[2]))}_s{c[3]}_r{int(c[4])}")
def test_pointwise_unit_matches_float64_reference(case):
    M, K, Cs, slack, relu = case
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M * 7 + K)
    Nn = sum(Cs)
    mode = N.lib().vt_pw_supported(N.VT_BF16, K, Cs[0], Cs[1] if len(Cs) > 1 else 0)
    assert mode in (1, 2)
    x = _rows(M, K, slack, gen)
    ws = [(torch.randn(c, K, device="cuda", generator=gen) * (2.0 / K) ** 0.5).to(BF) for c in Cs]
    d = _desc(x, ws, relu)
    st = stream()
    lib = N.lib()
    W = torch.cat(ws, 0).double()
    zb = (x.double() @ W.T).to(BF).double()  # [M][N], rounded where the unfused path stores z

    # ---- forward statistics -----------------------------------------------------------------------------------
    stats = [N.stats_buffer(c) for c in Cs]
    N.check(lib.vt_pw_fwd_stats(C.byref(d), _vps(stats), st))
    got = torch.cat([N.stats_decode(s) for s in stats], 1)  # [2][N]
    ref = torch.stack([zb.sum(0), (zb * zb).sum(0)])
    assert ((got - ref).abs() / ref.abs().max(1, keepdim=True).values).max().item() < 1e-3

    # ---- forward normalise (+ residual) ---------------------------------------------------------------------------
    mean = zb.mean(0)
    var = zb.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = torch.rand(Nn, device="cuda", generator=gen).double() + 0.5
    beta = torch.randn(Nn, device="cuda", generator=gen).double() * 0.3
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    coef = torch.stack([scale, shift, mean, invstd]).float().contiguous()
    scale, shift, mean, invstd = [c.double() for c in coef]  # what the kernels read
    ys = [_rows(M, c, slack, gen) for c in Cs]
    with_res = (M % 3) != 0  # (a residual for every group or for none)
    res = [_rows(M, c, slack, gen, 0.5) if with_res else None for c in Cs]
    N.check(lib.vt_pw_fwd_apply(C.byref(d), coef.data_ptr(), _vps(ys), _arr(C.c_int32, [y.stride(0) for y in ys]),
                                _vps(res), _arr(C.c_int32, [r.stride(0) if r is not None else 0 for r in res]), st))
    pre = zb * scale + shift
    yref = torch.relu(pre) if relu else pre
    off = 0
    for g, c in enumerate(Cs):
        r = yref[:, off: off + c] + (res[g].double() if res[g] is not None else 0.0)
        assert rel_err(ys[g], r) < 4e-3
        assert ((ys[g].double() - r).abs() / (r.abs() + 1.0)).max().item() < 0.05
        off += c

    # ---- backward reduction -------------------------------------------------------------------------------------
    dys = [_rows(M, c, slack, gen) for c in Cs]
    dy = torch.cat([t.double() for t in dys], 1)
    g_ = dy * (pre > 0) if relu else dy
    sums = [N.stats_buffer(c) for c in Cs]
    N.check(lib.vt_pw_bwd_reduce(C.byref(d), coef.data_ptr(), _vps(dys), _arr(C.c_int32, [t.stride(0) for t in dys]),
                                 _vps(sums), st))
    got = torch.cat([N.stats_decode(s) for s in sums], 1)
    ref = torch.stack([g_.sum(0), (g_ * (zb - mean) * invstd).sum(0)])
    # (a z within rounding of the ReLU threshold may fall on the other side of the mask than in float64)
    assert ((got - ref).abs() / ref.abs().max(1, keepdim=True).values.clamp_min(1.0)).max().item() < 3e-3

    # ---- backward apply: dx (+ addend), dW or dz ------------------------------------------------------------------
    bcoefs, off = [], 0
    a_, b_, d_ = scale, torch.randn(Nn, device="cuda", generator=gen).double() * 0.05, torch.randn(Nn, device="cuda", generator=gen).double() * 0.05
    for c in Cs:
        bcoefs.append(torch.stack([a_[off: off + c], b_[off: off + c], d_[off: off + c]]).float().contiguous())
        off += c
    a_ = torch.cat([b[0] for b in bcoefs]).double()
    b_ = torch.cat([b[1] for b in bcoefs]).double()
    d_ = torch.cat([b[2] for b in bcoefs]).double()
    dzref = (a_ * g_ - b_ * zb + d_).to(BF).double()
    add = _rows(M, K, slack, gen, 0.5)
    dx = _rows(M, K, slack, gen)
    dws = [torch.randn(c, K, device="cuda", generator=gen) for c in Cs] if mode == 2 else [None] * len(Cs)
    dw0 = [t.clone() if t is not None else None for t in dws]
    dzs = [_rows(M, c, slack, gen) for c in Cs] if mode == 1 else [None] * len(Cs)
    N.check(lib.vt_pw_bwd_apply(C.byref(d), coef.data_ptr(), _vps(dys), _arr(C.c_int32, [t.stride(0) for t in dys]),
                                _vps(bcoefs), dx.data_ptr(), dx.stride(0), add.data_ptr(), add.stride(0),
                                _vps(dws), _arr(C.c_int32, [K] * len(Cs)),
                                _vps(dzs), _arr(C.c_int32, [t.stride(0) if t is not None else 0 for t in dzs]), st))
    dxref = dzref @ W + add.double()
    assert rel_err(dx, dxref) < 4e-3
    off = 0
    for g, c in enumerate(Cs):
        if mode == 2:
            dwref = dzref[:, off: off + c].T @ x.double()
            assert rel_err(dws[g].double() - dw0[g].double(), dwref) < 2e-3
        else:
            assert rel_err(dzs[g], dzref[:, off: off + c]) < 4e-3
        off += c

    # ---- in place: the addend may BE the destination (the engine accumulates into an existing gradient) -----------
    dx2 = add.clone()
    N.check(lib.vt_pw_bwd_apply(C.byref(d), coef.data_ptr(), _vps(dys), _arr(C.c_int32, [t.stride(0) for t in dys]),
                                _vps(bcoefs), dx2.data_ptr(), dx2.stride(0), dx2.data_ptr(), dx2.stride(0),
                                _vps([None] * len(Cs)), _arr(C.c_int32, [K] * len(Cs)),
                                _vps(dzs), _arr(C.c_int32, [t.stride(0) if t is not None else 0 for t in dzs]), st))
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx if slack == 0 else dx.contiguous())


def test_pointwise_passes_agree_with_the_unfused_kernels():
    """the same unit through vt_conv_igemm + vt_bn_* (z and dz materialised): statistics equal to 1e-5, y and dx to one
    bf16 ulp on a handful of elements (different k order inside z), dW to 1e-3"""
    from gpu_util import conv_desc, vp

    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    B, H, Wd, K, Nn = 4, 24, 24, 64, 64
    M = B * H * Wd
    x4 = (torch.randn(B, H, Wd, K, device="cuda", generator=gen)).to(BF)
    w = (torch.randn(Nn, K, device="cuda", generator=gen) * (2.0 / K) ** 0.5).to(BF)
    lib, st = N.lib(), stream()
    # unfused
    z = torch.empty(B, H, Wd, Nn, device="cuda", dtype=BF)
    stats_u = N.stats_buffer(Nn)
    d = conv_desc(N.VT_BF16, x4, K, Nn, 1, 1, 0, Nn, N.VT_CONV_STATS)
    N.check(lib.vt_conv_igemm(C.byref(d), vp(x4), vp(w), vp(z), None, None, None, vp(stats_u), st))
    # fused
    x = x4.view(M, K)
    pd = _desc(x, [w], True)
    stats_f = N.stats_buffer(Nn)
    N.check(lib.vt_pw_fwd_stats(C.byref(pd), _vps([stats_f]), st))
    su, sf = N.stats_decode(stats_u), N.stats_decode(stats_f)
    assert ((su - sf).abs() / su.abs().max(1, keepdim=True).values).max().item() < 1e-4
    zb = z.view(M, Nn).double()
    mean = zb.mean(0)
    invstd = 1.0 / torch.sqrt(zb.var(0, unbiased=False) + 1e-5)
    coef = torch.stack([invstd, -mean * invstd, mean, invstd]).float().contiguous()
    y_u = torch.empty(M, Nn, device="cuda", dtype=BF)
    N.check(lib.vt_bn_act_apply(vp(z), Nn, coef[0].data_ptr(), coef[1].data_ptr(), None, 0, vp(y_u), Nn, M, Nn, 1, N.VT_BF16, st))
    y_f = torch.empty(M, Nn, device="cuda", dtype=BF)
    N.check(lib.vt_pw_fwd_apply(C.byref(pd), coef.data_ptr(), _vps([y_f]), _arr(C.c_int32, [Nn]), _vps([None]), _arr(C.c_int32, [0]), st))
    torch.cuda.synchronize()
    neq = (y_u != y_f).float().mean().item()
    assert neq < 0.02 and rel_err(y_f, y_u.double()) < 2e-3, neq


@pytest.mark.parametrize("M,K,slack,relu,with_res", [(1555, 80, 0, True, True), (2048, 80, 16, False, False),
                                                     (1111, 160, 0, True, False), (16 * 64, 160, 32, True, True)],
                         ids=lambda v: str(v))
def test_apply_pass_on_the_80_channel_shapes(M, K, slack, relu, with_res):
    """Round 4: the apply pass ALONE on channel counts that are no multiples of 32 -- YOLOv5x's first stage in inference
    (80 -> 80 bottleneck conv1, 160 -> 80 CSP conv1 / conv2: darknet.py:124-133 at width 1.25; components.py:26-44 with
    running statistics) -- against float64 on the same bf16 operands.  The kernel is 96 wide: filter rows / columns and
    coefficients beyond 80 do not exist, stores beyond them are masked and x / residual loads beyond them read the pixel's
    first bytes; operands sit in NaN-filled wider buffers (slices) and the outputs' neighbours must stay NaN."""
    Cout = 80
    lib = N.lib()
    assert lib.vt_pw_apply_supported(N.VT_BF16, K, Cout) == 1 and lib.vt_pw_supported(N.VT_BF16, K, Cout, 0) == 0
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    x = _rows(M, K, slack, gen)
    w = (torch.randn(Cout, K, device="cuda", generator=gen) * (2.0 / K) ** 0.5).to(BF)
    d = _desc(x, [w], relu)
    zb = (x.double() @ w.double().T).to(BF).double()
    scale = torch.rand(Cout, device="cuda", generator=gen) + 0.5
    shift = torch.randn(Cout, device="cuda", generator=gen) * 0.3
    coef = torch.stack([scale, shift, torch.zeros_like(scale), torch.ones_like(scale)]).contiguous()
    ywide = torch.full((M, Cout + 32), float("nan"), device="cuda", dtype=BF)
    y = ywide[:, 16: 16 + Cout]
    res = _rows(M, Cout, slack, gen, 0.5) if with_res else None
    N.check(lib.vt_pw_fwd_apply(C.byref(d), coef.data_ptr(), _vps([y]), _arr(C.c_int32, [y.stride(0)]), _vps([res]),
                                _arr(C.c_int32, [res.stride(0) if res is not None else 0]), stream()))
    torch.cuda.synchronize()
    pre = zb * scale.double() + shift.double()
    ref = (torch.relu(pre) if relu else pre) + (res.double() if res is not None else 0.0)
    assert not torch.isnan(y.float()).any()
    assert rel_err(y, ref) < 4e-3 and ((y.double() - ref).abs() / (ref.abs() + 1.0)).max().item() < 0.05
    assert torch.isnan(ywide[:, :16].float()).all() and torch.isnan(ywide[:, 16 + Cout:].float()).all()
    # the statistics / backward passes have no such shapes: they must refuse, not run a wrong kernel
    stats = N.stats_buffer(Cout)
    assert lib.vt_pw_fwd_stats(C.byref(d), _vps([stats]), stream()) == N.VT_ERR_UNSUPPORTED


FIN_CASES = [c for c in CASES if sum(c[2]) <= 128] + [(1555, 160, [160], 0, True)]  # (160 channels: the separate launches inside)


@pytest.mark.parametrize("case", FIN_CASES, ids=lambda c: f"M{c[0]}_K{c[1]}_C{'+'.join(map(str, c[2]))}_s{c[3]}_r{int(c[4])}")
def test_apply_passes_that_finalize_for_themselves_are_bit_identical(case):
    """vt_pw_fwd_apply_finalize / vt_pw_bwd_apply_finalize (round 6: every workgroup finalizes the BatchNorm coefficients in its
    prologue) against vt_bn_finalize + vt_pw_fwd_apply and vt_bn_bwd_finalize + vt_pw_bwd_apply per group: every output EQUAL
    (coefficients, running statistics, batch counters, y; d(gamma), d(beta), the backward coefficients, dx, dz; dW up to
    the order of its f32 atomics), one launch
    instead of 1 + ngroups."""
    M, K, Cs, slack, relu = case
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    G, Nn = len(Cs), sum(Cs)
    lib, st = N.lib(), stream()
    x = _rows(M, K, slack, gen)
    ws = [(torch.randn(c, K, device="cuda", generator=gen) * (2.0 / K) ** 0.5).to(BF) for c in Cs]
    pd = _desc(x, ws, relu)
    gammas = [torch.rand(c, device="cuda", generator=gen) + 0.5 for c in Cs]
    betas = [torch.randn(c, device="cuda", generator=gen) * 0.2 for c in Cs]
    ress = [_rows(M, c, 0, gen) for c in Cs]
    dys = [_rows(M, c, 0, gen) for c in Cs]
    add = _rows(M, K, 0, gen)
    stats = [N.stats_buffer(c) for c in Cs]
    N.check(lib.vt_pw_fwd_stats(C.byref(pd), _vps(stats), st))
    full = Nn * K <= 4096
    offs = [0, Cs[0]]
    ldy = _arr(C.c_int32, Cs)

    def run(fused):
        coef = torch.full((4, Nn), float("nan"), device="cuda")
        rms, rvs = [torch.full((c,), 0.1, device="cuda") for c in Cs], [torch.full((c,), 0.9, device="cuda") for c in Cs]
        nbts = [torch.full((1,), 3 + g, dtype=torch.int64, device="cuda") for g in range(G)]
        ys = [torch.full((M, c), float("nan"), device="cuda", dtype=BF) for c in Cs]
        before = N.launch_count()
        if fused:
            fin = (N.BnFinFwd * G)()
            for g in range(G):
                fin[g] = N.BnFinFwd(stats[g].data_ptr(), float(M), gammas[g].data_ptr(), betas[g].data_ptr(), 1e-5, 0.1, rms[g].data_ptr(),
                                    rvs[g].data_ptr(), nbts[g].data_ptr())
            N.check(lib.vt_pw_fwd_apply_finalize(C.byref(pd), fin, coef.data_ptr(), _vps(ys), ldy, _vps(ress), ldy, st))
        else:
            for g in range(G):
                rows = [coef[i, offs[g]:offs[g] + Cs[g]].data_ptr() for i in range(4)]
                N.check(lib.vt_bn_finalize(stats[g].data_ptr(), Cs[g], float(M), gammas[g].data_ptr(), betas[g].data_ptr(), 1e-5, 0.1,
                                           rms[g].data_ptr(), rvs[g].data_ptr(), nbts[g].data_ptr(), *rows, st))
            N.check(lib.vt_pw_fwd_apply(C.byref(pd), coef.data_ptr(), _vps(ys), ldy, _vps(ress), ldy, st))
        torch.cuda.synchronize()
        n_fwd = N.launch_count() - before
        # backward on the coefficients just made
        sums = [N.stats_buffer(c) for c in Cs]
        N.check(lib.vt_pw_bwd_reduce(C.byref(pd), coef.data_ptr(), _vps(dys), ldy, _vps(sums), st))
        dgs, dbs = [torch.full((c,), 0.25, device="cuda") for c in Cs], [torch.full((c,), -0.5, device="cuda") for c in Cs]
        bcoefs = [torch.full((3, c), float("nan"), device="cuda") for c in Cs]
        dx = torch.full((M, K), float("nan"), device="cuda", dtype=BF)
        dws = [torch.full((c, K), 0.5, device="cuda") if full else None for c in Cs]
        dzs = [None if full else torch.full((M, c), float("nan"), device="cuda", dtype=BF) for c in Cs]
        tail = (dx.data_ptr(), K, add.data_ptr(), K, _vps(dws), _arr(C.c_int32, [K] * G), _vps(dzs), ldy, st)
        before = N.launch_count()
        if fused:
            bfin = (N.BnFinBwd * G)()
            for g in range(G):
                bfin[g] = N.BnFinBwd(sums[g].data_ptr(), float(M), 0.5, 1, dgs[g].data_ptr(), dbs[g].data_ptr())
            N.check(lib.vt_pw_bwd_apply_finalize(C.byref(pd), coef.data_ptr(), _vps(dys), ldy, bfin, _vps(bcoefs), *tail))
        else:
            for g in range(G):
                sl = slice(offs[g], offs[g] + Cs[g])
                N.check(lib.vt_bn_bwd_finalize(sums[g].data_ptr(), Cs[g], float(M), 0.5, coef[0, sl].data_ptr(), coef[2, sl].data_ptr(),
                                               coef[3, sl].data_ptr(), 1, dgs[g].data_ptr(), dbs[g].data_ptr(), bcoefs[g].data_ptr(), st))
            N.check(lib.vt_pw_bwd_apply(C.byref(pd), coef.data_ptr(), _vps(dys), ldy, _vps(bcoefs), *tail))
        torch.cuda.synchronize()
        n_bwd = N.launch_count() - before
        outs = [coef, *rms, *rvs, *nbts, *ys, *dgs, *dbs, *bcoefs, dx, *[t for t in dzs if t is not None]]
        return outs, [t for t in dws if t is not None], n_fwd, n_bwd

    ref, dw0, nf0, nb0 = run(False)
    got, dw1, nf1, nb1 = run(True)
    for a, b in zip(dw1, dw0):  # (f32 atomics across workgroups: equal up to the order of the partial sums)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * b.abs().max().item())
    assert (nf0, nb0) == (1 + G, 1 + G)
    assert (nf1, nb1) == ((1, 1) if Nn <= 128 else (1 + G, 1 + G))
    for a, b in zip(got, ref):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())) and torch.equal(torch.isnan(a.float()), torch.isnan(b.float()))
