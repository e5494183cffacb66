"""vt_wgrad6.hip: the CU-owning filter-gradient kernel of the stride-1 3x3 bf16 layers (autograd backward of the
nn.Conv2d inside ConvNormAct, reference components.py:26-35, w.r.t. its weight), one layer per launch and several
same-shape layers per launch (vt_conv_wgrad_group), against torch's float64 conv2d backward on the SAME bf16-rounded
operands.  Tolerance 2e-5 relative L2: the products are exact in f32, only the summation order differs.

Cases: full and ragged channel tiles (72, 136, 160, 192 channels), maps from 5 x 7 to 57 x 41 (padded pitch below and
above the 64-position step, odd sizes), channel slices of wider buffers (pixel stride > channels), pixel counts that end
in a partial step, and the kernel forced / forbidden through its knob (the old all-taps kernel is the comparison)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vision_toolbox import _native as N

from gpu_util import conv_desc, rel_err, stream, vp

pytestmark = pytest.mark.gpu

# B, Cin, Cout, H, W
CASES = [
    (3, 64, 64, 28, 28),
    (2, 128, 128, 14, 14),
    (5, 72, 136, 7, 7),
    (2, 160, 192, 9, 13),
    (1, 64, 128, 57, 41),
    (4, 40, 48, 5, 7),
    (2, 224, 224, 7, 7),
    (1, 128, 64, 112, 112),
]


def _operands(B, Cin, Cout, H, W, seed, ldx=None, ldy=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).to(torch.bfloat16)
    dz = torch.randn(B, Cout, H, W, generator=g).to(torch.bfloat16)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(dz.double())
    ref = w.grad.permute(0, 2, 3, 1).contiguous()  # [Cout][kh][kw][Cin]

    def dev(t, ld, coff):
        v = t.permute(0, 2, 3, 1).contiguous().cuda()
        if ld is None:
            return v
        wide = torch.full((*v.shape[:3], ld), float("nan"), device="cuda", dtype=torch.bfloat16)
        wide[..., coff:coff + v.shape[3]] = v
        return wide[..., coff:coff + v.shape[3]]

    return dev(x, ldx, 8), dev(dz, ldy, 16), ref


@pytest.fixture(autouse=True)
def _knob():
    N.lib()
    yield
    N.check(N.lib().vt_set_knob(b"VT_WGRAD6", 1))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_single_layer_against_float64(case):
    B, Cin, Cout, H, W = case
    xd, dzd, ref = _operands(B, Cin, Cout, H, W, seed=sum(case))
    d = conv_desc(N.VT_BF16, xd, Cin, Cout, 3, 1, 1, Cout)
    K = 9 * Cin
    dw = torch.full((Cout, K), 0.25, device="cuda")  # the kernel accumulates into the existing gradient
    N.check(N.lib().vt_conv_wgrad(C.byref(d), vp(xd), vp(dzd), vp(dw), K, stream()))
    torch.cuda.synchronize()
    assert N.last_kernel_name().startswith("wgrad6_kernel"), N.last_kernel_name()
    got = (dw - 0.25).view(Cout, 3, 3, Cin).cpu()
    assert rel_err(got, ref) < 2e-5
    # and the kernel it replaces, on the same operands (both against float64; never bitwise: the pixel splits differ)
    N.check(N.lib().vt_set_knob(b"VT_WGRAD6", 0))
    dw2 = torch.zeros(Cout, K, device="cuda")
    N.check(N.lib().vt_conv_wgrad(C.byref(d), vp(xd), vp(dzd), vp(dw2), K, stream()))
    torch.cuda.synchronize()
    assert not N.last_kernel_name().startswith("wgrad6_kernel")
    assert rel_err(dw2.view(Cout, 3, 3, Cin).cpu(), ref) < 2e-5


def test_channel_slices_of_wider_buffers():
    B, Cin, Cout, H, W = 2, 96, 80, 12, 20
    xd, dzd, ref = _operands(B, Cin, Cout, H, W, seed=7, ldx=Cin + 40, ldy=Cout + 24)
    d = conv_desc(N.VT_BF16, xd, Cin, Cout, 3, 1, 1, Cout + 24)
    assert d.ldx == Cin + 40
    K = 9 * Cin
    ldgw = K + 16  # a gradient row pitch wider than the filter row
    dw = torch.zeros(Cout, ldgw, device="cuda")
    N.check(N.lib().vt_conv_wgrad(C.byref(d), vp(xd), vp(dzd), vp(dw), ldgw, stream()))
    torch.cuda.synchronize()
    assert N.last_kernel_name().startswith("wgrad6_kernel")
    assert rel_err(dw[:, :K].reshape(Cout, 3, 3, Cin).cpu(), ref) < 2e-5
    assert torch.count_nonzero(dw[:, K:]).item() == 0


@pytest.mark.parametrize("n,shape", [(3, (2, 128, 128, 14, 14)), (8, (1, 64, 64, 28, 28)), (11, (1, 72, 64, 7, 9)),
                                     (2, (2, 32, 32, 16, 16))], ids=str)
def test_grouped_launch_equals_the_layers_one_by_one(n, shape):
    """vt_conv_wgrad_group: n same-shape layers; more than 8 are cut into launches of <= 8; a shape the CU-owning kernel
    does not cover (32 channels) runs layer by layer through the same entry point."""
    B, Cin, Cout, H, W = shape
    ops = [_operands(B, Cin, Cout, H, W, seed=100 + i) for i in range(n)]
    d = conv_desc(N.VT_BF16, ops[0][0], Cin, Cout, 3, 1, 1, Cout)
    K = 9 * Cin
    dws = [torch.full((Cout, K), float(i), device="cuda") for i in range(n)]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    N.check(N.lib().vt_conv_wgrad_group(C.byref(d), n, arr([o[0] for o in ops]), arr([o[1] for o in ops]), arr(dws), K,
                                        stream()))
    torch.cuda.synchronize()
    if Cin > 32:
        assert N.last_kernel_name().startswith("wgrad6_kernel"), N.last_kernel_name()
    for i in range(n):
        got = (dws[i] - float(i)).view(Cout, 3, 3, Cin).cpu()
        assert rel_err(got, ops[i][2]) < 2e-5, i


def test_group_entry_rejects_bad_arguments():
    xd, dzd, _ = _operands(1, 64, 64, 8, 8, seed=1)
    d = conv_desc(N.VT_BF16, xd, 64, 64, 3, 1, 1, 64)
    dw = torch.zeros(64, 9 * 64, device="cuda")
    one = lambda t: (C.c_void_p * 1)(t.data_ptr())
    assert N.lib().vt_conv_wgrad_group(C.byref(d), 0, one(xd), one(dzd), one(dw), 9 * 64, stream()) == N.VT_ERR_INVALID
    null = (C.c_void_p * 1)(None)
    assert N.lib().vt_conv_wgrad_group(C.byref(d), 1, null, one(dzd), one(dw), 9 * 64, stream()) == N.VT_ERR_INVALID
    N.check(N.lib().vt_memset(vp(dw), 0, 16, stream()))


@pytest.mark.parametrize("n,shape", [(8, (4, 128, 128, 28, 28)), (5, (3, 256, 192, 14, 14)), (2, (2, 72, 40, 9, 7))], ids=str)
def test_grouped_launch_of_1x1_layers_on_the_general_kernel(n, shape):
    """vt_conv_wgrad_group on 1x1 stride-1 layers (DarknetBlock.conv1, reference darknet.py:23): one launch of the general
    kernel over the group, each layer against the float64 product of its own bf16 operands, accumulated into the existing
    gradient; channel counts that are not multiples of the 128-wide tiles included."""
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(n * 31 + Cin)
    xs = [torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda() for _ in range(n)]
    dzs = [torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).cuda() for _ in range(n)]
    d = conv_desc(N.VT_BF16, xs[0], Cin, Cout, 1, 1, 0, Cout)
    dws = [torch.full((Cout, Cin), float(i), device="cuda") for i in range(n)]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    before = N.launch_count()
    N.set_knob("VT_WGRAD_GROUP_1X1", 1)  # (off by default: NOTEBOOK R5.18)
    try:
        N.check(N.lib().vt_conv_wgrad_group(C.byref(d), n, arr(xs), arr(dzs), arr(dws), Cin, stream()))
        torch.cuda.synchronize()
    finally:
        N.set_knob("VT_WGRAD_GROUP_1X1", 0)
    assert N.launch_count() - before == 1, "one launch for the group"
    assert N.last_kernel_name().startswith("wgrad_kernel"), N.last_kernel_name()
    for i in range(n):
        ref = dzs[i].double().reshape(-1, Cout).t() @ xs[i].double().reshape(-1, Cin)
        assert rel_err((dws[i] - float(i)).cpu(), ref.cpu()) < 2e-5, i
