"""The BatchNorm finalize step as the tail of the launch that produces its sums (round 6, csrc/vt_fin_tail.h):
vt_bn_act_bwd_reduce_finalize against vt_bn_act_bwd_reduce + vt_bn_bwd_finalize (autograd backward of reference
components.py:36-44; the separate calls are pinned to the oracle in test_kernels_gpu.py).

  * the sums are exact integers and the finalize arithmetic is one shared function: d(gamma), d(beta), the three coefficient
    rows and the sums themselves are BIT-IDENTICAL to the two launches, whichever workgroup drew the last ticket -- checked
    over repeated launches (another arrival order each time);
  * one launch where the tail fits (<= 128 channels per channel group of the reduction's grid), exactly two otherwise and
    with the knob VT_FIN_TAIL = 0 or tickets == NULL;
  * the ticket words are zero again after the call."""
import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# M, C, launches
SHAPES = [
    (256 * 28 * 28, 128, 1),   # one channel group, 8 passes of 16 channels
    (256 * 14 * 14, 256, 1),   # 4 channel groups of 64 (M <= 65536): a ticket per group
    (256 * 7 * 7, 512, 1),
    (256 * 7 * 7, 1024, 1),    # 16 channel groups
    (256 * 56 * 56, 64, 1),
    (3001, 32, 1),
    (77, 8, 1),                # fewer rows than one workgroup pass: a grid of one workgroup
    (256 * 28 * 28, 256, 2),   # one channel group of 256 channels: four batches, not offered
    (5000, 160, 2),
]


def _inputs(M, Cc):
    torch.manual_seed(M % 997 + Cc)
    dy = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
    z = (torch.randn(M, Cc, device="cuda") * 1.3 + 0.4).to(torch.bfloat16)
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.3
    mean = z.float().mean(0)
    invstd = 1.0 / torch.sqrt(z.float().var(0, unbiased=False) + 1e-5)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    return dy, z, scale, shift, mean.contiguous(), invstd.contiguous()


def _outs(Cc):
    return (N.stats_buffer(Cc), torch.full((Cc,), 0.25, device="cuda"), torch.full((Cc,), -0.5, device="cuda"),
            torch.full((3, Cc), float("nan"), device="cuda"))


def _two(ops, M, Cc, relu, train):
    dy, z, scale, shift, mean, invstd = ops
    lib = N.lib()
    sums, dgamma, dbeta, coef = _outs(Cc)
    N.check(lib.vt_bn_act_bwd_reduce(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, relu, N.VT_BF16,
                                     vp(sums), stream()))
    N.check(lib.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 0.5, vp(scale), vp(mean), vp(invstd), train, vp(dgamma), vp(dbeta),
                                   vp(coef), stream()))
    torch.cuda.synchronize()
    return sums, dgamma, dbeta, coef


def _one(ops, M, Cc, relu, train, tickets):
    dy, z, scale, shift, mean, invstd = ops
    lib = N.lib()
    sums, dgamma, dbeta, coef = _outs(Cc)
    before = N.launch_count()
    N.check(lib.vt_bn_act_bwd_reduce_finalize(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, relu,
                                              N.VT_BF16, vp(sums), float(M), 0.5, train, vp(dgamma), vp(dbeta), vp(coef),
                                              vp(tickets) if tickets is not None else None, stream()))
    torch.cuda.synchronize()
    return (sums, dgamma, dbeta, coef), N.launch_count() - before


@pytest.mark.parametrize("relu,train", [(1, 1), (0, 1), (3, 0)], ids=["relu", "no_act", "silu_frozen_stats"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_tail_finalize_is_bit_identical_to_the_two_launches(shape, relu, train):
    M, Cc, launches = shape
    ops = _inputs(M, Cc)
    ref = _two(ops, M, Cc, relu, train)
    tickets = torch.zeros(N.VT_FIN_TICKETS, dtype=torch.int32, device="cuda")
    for rep in range(6 if launches == 1 else 1):
        got, n = _one(ops, M, Cc, relu, train, tickets)
        assert n == launches, (n, N.last_kernel_name())
        assert int(tickets.abs().sum().item()) == 0
        for a, b in zip(got, ref):
            assert torch.equal(a, b), rep


def test_tail_finalize_knob_and_null_tickets_fall_back_to_two_launches():
    M, Cc = 4096, 64
    ops = _inputs(M, Cc)
    ref = _two(ops, M, Cc, 1, 1)
    got, n = _one(ops, M, Cc, 1, 1, None)
    assert n == 2 and all(torch.equal(a, b) for a, b in zip(got, ref))
    tickets = torch.zeros(N.VT_FIN_TICKETS, dtype=torch.int32, device="cuda")
    N.set_knob("VT_FIN_TAIL", 0)
    try:
        got, n = _one(ops, M, Cc, 1, 1, tickets)
    finally:
        N.set_knob("VT_FIN_TAIL", 1)
    assert n == 2 and all(torch.equal(a, b) for a, b in zip(got, ref))


# ---- forward: vt_conv_igemm_finalize against vt_conv_igemm(STATS) + vt_bn_finalize ---------------------------------------
# B, Cin, Cout, H, W, launches (1: the persistent two-group kernel takes the launch and runs the tail)
CONV_SHAPES = [
    (128, 128, 128, 28, 28, 1),   # the dominant layer's geometry (padded rows)
    (256, 256, 256, 14, 14, 1),   # two filter tiles, K split: group 1 hands its accumulators over before the epilogue
    (40, 64, 128, 45, 37, 1),     # odd sizes, M not a multiple of 32
    (20, 64, 256, 41, 52, 1),     # two filter tiles
    (33, 64, 128, 29, 71, 1),
    (12, 64, 384, 41, 52, 2),     # 384 channels: more than the tail covers
    (256, 32, 128, 28, 28, 2),    # another kernel takes the launch: the finalize step is a launch of its own
    (4, 16, 32, 9, 9, 2),
]


def _conv_desc(B, Cin, Cout, H, W):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, Cin
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, W, 1, 1, -1, -1
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, Cout, H, W, 1, 1
    d.ldw, d.ldr, d.flags, d.ntaps = 9 * Cin, 0, N.VT_CONV_STATS, 9
    for i in range(9):
        d.dh[i], d.dw[i] = i // 3, i % 3
    return d


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv_tail_finalize_is_bit_identical_to_the_two_launches(shape):
    import ctypes as C
    B, Cin, Cout, H, W, launches = shape
    torch.manual_seed(B + Cin + H)
    lib = N.lib()
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.3
    d = _conv_desc(B, Cin, Cout, H, W)
    count = float(B * H * W)

    def outs():
        return (torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16), N.stats_buffer(Cout),
                torch.full((Cout,), 0.1, device="cuda"), torch.full((Cout,), 0.9, device="cuda"),
                torch.full((1,), 7, device="cuda", dtype=torch.int64), torch.full((4, Cout), float("nan"), device="cuda"))

    z, st, rm, rv, nbt, coef = ref = outs()
    N.check(lib.vt_conv_igemm(C.byref(d), vp(x), vp(w), vp(z), None, None, None, vp(st), stream()))
    kernel = N.last_kernel_name()
    N.check(lib.vt_bn_finalize(vp(st), Cout, count, vp(gamma), vp(beta), 1e-5, 0.1, vp(rm), vp(rv), vp(nbt), vp(coef[0]), vp(coef[1]),
                               vp(coef[2]), vp(coef[3]), stream()))
    torch.cuda.synchronize()
    assert nbt.item() == 8 and (kernel.startswith("span6") or launches == 2), kernel
    tickets = torch.zeros(N.VT_FIN_TICKETS, dtype=torch.int32, device="cuda")
    for rep in range(5 if launches == 1 else 1):
        z, st, rm, rv, nbt, coef = got = outs()
        before = N.launch_count()
        N.check(lib.vt_conv_igemm_finalize(C.byref(d), vp(x), vp(w), vp(z), vp(st), count, vp(gamma), vp(beta), 1e-5, 0.1, vp(rm), vp(rv),
                                           vp(nbt), vp(coef[0]), vp(coef[1]), vp(coef[2]), vp(coef[3]), vp(tickets), stream()))
        torch.cuda.synchronize()
        assert N.launch_count() - before == launches, (N.launch_count() - before, N.last_kernel_name())
        assert int(tickets.abs().sum().item()) == 0
        for a, b in zip(got, ref):
            assert torch.equal(a, b), rep
    # the inference statistics moved and the coefficients are those of the batch (float64 on the stored z)
    zs = ref[0].double().reshape(-1, Cout)
    torch.testing.assert_close(ref[5][2].double(), zs.mean(0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ref[2].double(), 0.9 * 0.1 + 0.1 * zs.mean(0), rtol=1e-5, atol=1e-6)
