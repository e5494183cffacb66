"""vt_igemm_span6.hip (the two-group + loader-wave kernel of the MFMA-bound 3x3 stride-1 layers: ConvNormAct
`components.py:26-35` inside DarknetBlock / CSP / OSA stages, forward and stride-1 data gradient) against
the kernel the dispatcher would otherwise take, on identical operands.  Against vt_igemm_span.hip the comparison is
BIT-EXACT (same products, same order); where the alternative is the general kernel (few output pixels) the two
differ by summation order only and are compared at one bf16 rounding.  Those kernels are pinned to the float64
oracle in test_kernels_gpu.py / test_fullsize_gpu.py.

The model-level GPU tests run toy sizes that never reach this kernel (it needs >= 65,536 output pixels x filter
tiles), so the dispatch is forced here (VT_SPAN6=2) over a matrix chosen for its edge cases: pixel counts that are
not multiples of the 32-row unit, odd map sizes (padded coordinates (H+1) x (W+1)), 6 / 7 piece taps (map widths 29
.. 80), filter-tile counts that do not divide 32 (3 tiles), a partial last filter tile (Cout = 160), channel-slice
operands (ldx > Cin, ldy > Cout: the concat-elided buffers of CSP / OSA blocks), every epilogue mode."""
import ctypes as C
import os

import pytest
import torch

from vision_toolbox import _native as N

from gpu_util import stream, vp

pytestmark = pytest.mark.gpu

# B, Cin, Cout, H, W
SHAPES = [
    (8, 64, 128, 96, 96),     # 7 piece taps (Wp = 97, the widest map whose span fits the LDS), 4..5-unit tiles
    (40, 64, 128, 45, 37),    # odd sizes, M = 66,600 (not a multiple of 32)
    (16, 96, 160, 64, 80),    # 7 piece taps (Wp = 81), partial second filter tile, three channel chunks
    (96, 128, 128, 28, 28),   # the dominant layer's geometry at 3/8 of the batch
    (12, 64, 384, 41, 52),    # three filter tiles (do not divide the 32 workgroups of an XCD)
    (128, 160, 160, 28, 28),  # VoVNet-39 stage 2
    (33, 64, 256, 29, 71),    # odd batch, two filter tiles
    (6, 64, 128, 112, 112),   # a map wider than 96: the tile height drops to 6 units so that the span still fits (VoVNet-39 stem)
    (8, 64, 128, 60, 140),    # ... and to 5 units at Wp = 141
    (256, 32, 128, 28, 28),   # a single channel chunk: NOT taken by span6 (must fall through unharmed)
]
MODES = [("stats", N.VT_CONV_STATS), ("plain", 0), ("residual", N.VT_CONV_RESIDUAL),
         ("affine_relu", N.VT_CONV_AFFINE | N.VT_CONV_RELU),
         ("affine_relu_residual", N.VT_CONV_AFFINE | N.VT_CONV_RELU | N.VT_CONV_RESIDUAL)]


def _desc(B, Cin, Cout, H, W, ldx, ldy, ldr, flags, flip):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, ldx
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, W, 1, 1, -1, -1
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, ldy, H, W, 1, 1
    d.ldw, d.ldr, d.flags, d.ntaps = 9 * Cin, ldr, flags, 9
    for i in range(9):
        # flip: the tap order of a stride-1 data gradient (filter rotated by 180 degrees)
        t = 8 - i if flip else i
        d.dh[i], d.dw[i] = t // 3, t % 3
    return d


def _run(env, d, x, w, y, scale, shift, res, stats):
    N.set_knob("VT_SPAN6", int(env))  # (the dispatcher reads the environment once per process; tests set the switch)
    try:
        N.check(N.lib().vt_conv_igemm(C.byref(d), vp(x), vp(w), vp(y), vp(scale) if scale is not None else None,
                                      vp(shift) if shift is not None else None, vp(res) if res is not None else None,
                                      vp(stats) if stats is not None else None, stream()))
        torch.cuda.synchronize()
        return N.last_kernel_name()
    finally:
        N.set_knob("VT_SPAN6", 1)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_span6_is_bit_identical_to_the_span_kernel(shape, mode):
    B, Cin, Cout, H, W = shape
    flags = mode[1]
    torch.manual_seed(hash(shape) % 1000)
    slices = shape[0] % 2 == 0  # half of the shapes: operands are channel slices of wider buffers
    ldx, ldy = (Cin + 32, Cout + 64) if slices else (Cin, Cout)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin] if slices else xb
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=mode[0] == "residual")
    outs = {}
    for env in ("0", "2"):
        yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
        y = yb[..., 32:32 + Cout] if slices else yb
        st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
        name = _run(env, d, x, w, y, scale, shift, res, st)
        outs[env] = (yb, N.stats_decode(st) if st is not None else None, name)
    (y0, s0, n0), (y1, s1, n1) = outs["0"], outs["2"]
    assert "span6" not in n0
    assert ("span6" in n1) == (Cin >= 64), n1  # the forced dispatch really took the kernel under test
    assert torch.equal(torch.isnan(y0.float()), torch.isnan(y1.float()))  # nothing outside the slice was written
    a, b = torch.nan_to_num(y0.float()), torch.nan_to_num(y1.float())
    if "span_kernel" in n0 or "span6" not in n1:
        assert torch.equal(a, b)
    else:  # reference = the general kernel: another summation order, one bf16 rounding apart at most
        assert ((a - b).norm() / a.norm()).item() < 1e-3
        torch.testing.assert_close(b, a, rtol=2.0 ** -6, atol=2e-2)  # two roundings (epilogue, residual add): 2 ulp
    if s0 is not None:
        # the statistics are sums of the (same) stored values in a different order
        torch.testing.assert_close(s1, s0, rtol=2e-3 if "span_kernel" not in n0 else 1e-5, atol=0.5)


@pytest.mark.parametrize("mode", [m for m in MODES if m[0] != "stats"], ids=[m[0] for m in MODES if m[0] != "stats"])
def test_160_columns_run_as_128_on_span6_plus_32_on_the_span_kernel(mode):
    """Cout = 128 k + 32 without batch statistics (Darknet-YOLOv5x's 160-channel layers; VoVNet-39's data gradients): two
    launches over the same input, columns [0, 128) on span6 and [128, 160) on the input-span kernel's 32-wide tile
    (VT_SPAN6_SPLIT).  Same values as the unsplit launch, nothing outside the channel slice."""
    B, Cin, Cout, H, W = 16, 96, 160, 64, 80
    flags = mode[1]
    torch.manual_seed(11)
    ldx, ldy = Cin + 32, Cout + 64
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin]
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=False)
    outs, launches = [], []
    try:
        for split in (2, 0):  # (2: the inference epilogues too; by default only the data gradients are split)
            N.set_knob("VT_SPAN6_SPLIT", split)
            yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
            before = N.launch_count()
            name = _run("2", d, x, w, yb[..., 32:32 + Cout], scale, shift, res, None)
            launches.append(N.launch_count() - before)
            assert "span6" in name
            outs.append(yb)
    finally:
        N.set_knob("VT_SPAN6_SPLIT", 1)
    assert launches == [2, 1], launches
    assert torch.equal(torch.isnan(outs[0].float()), torch.isnan(outs[1].float()))
    assert torch.equal(torch.nan_to_num(outs[0].float()), torch.nan_to_num(outs[1].float()))


@pytest.mark.parametrize("mode", [m for m in MODES if m[0] != "stats"], ids=[m[0] for m in MODES if m[0] != "stats"])
def test_160_columns_tail_on_the_persistent_span_kernel(mode):
    """Round 4: from 262144 rows up the 32-column tail of a 128 k + 32 column launch runs on vt_igemm_pspan.hip (default
    dispatch, forward epilogues included: Darknet-YOLOv5x's 160 -> 160 @80x80 layers).  Same products in the same (chunk,
    tap) order as the unsplit launch; compared within two bf16 roundings, nothing outside the channel slice."""
    B, Cin, Cout, H, W = 16, 96, 160, 128, 128
    flags = mode[1]
    torch.manual_seed(12)
    ldx, ldy = Cin + 32, Cout + 64
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin]
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=False)
    outs, launches = [], []
    try:
        for split in (1, 0):
            N.set_knob("VT_SPAN6_SPLIT", split)
            yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
            before = N.launch_count()
            name = _run("1", d, x, w, yb[..., 32:32 + Cout], scale, shift, res, None)
            launches.append(N.launch_count() - before)
            assert "span6" in name
            outs.append(yb)
    finally:
        N.set_knob("VT_SPAN6_SPLIT", 1)
    assert launches == [2, 1], launches
    assert torch.equal(torch.isnan(outs[0].float()), torch.isnan(outs[1].float()))
    a, b = torch.nan_to_num(outs[0].float()), torch.nan_to_num(outs[1].float())
    torch.testing.assert_close(a, b, rtol=2.0 ** -6, atol=2e-2)
    assert torch.equal(a[..., 32:32 + 128], b[..., 32:32 + 128])  # the head columns: the same kernel, the same bits


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", [s for s in SHAPES if s[1] >= 64], ids=lambda s: "x".join(map(str, s)))
def test_masked_rows_are_bit_identical_to_padded_positions(shape, mode):
    """Round 5: the MASKED form of span6 (rows = pixels, a tap outside the image reads a zero block) against the padded
    form (rows = positions of the (H+1) x (W+1) image) on identical operands: the masked taps contribute exact zeros, the
    summation order is the same -- bit for bit, statistics included."""
    B, Cin, Cout, H, W = shape
    flags = mode[1]
    torch.manual_seed(hash(shape) % 1000 + 1)
    slices = shape[0] % 2 == 1
    ldx, ldy = (Cin + 32, Cout + 64) if slices else (Cin, Cout)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin] if slices else xb
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=mode[0] == "residual")
    outs = []
    try:
        for mask in (0, 2):
            N.set_knob("VT_SPAN6_MASK", mask)
            yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
            y = yb[..., 32:32 + Cout] if slices else yb
            st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
            name = _run("2", d, x, w, y, scale, shift, res, st)
            outs.append((yb, N.stats_decode(st) if st is not None else None, name))
    finally:
        N.set_knob("VT_SPAN6_MASK", 1)
    (y0, s0, n0), (y1, s1, n1) = outs
    assert "span6" in n0 and "masked" not in n0 and "masked" in n1, (n0, n1)
    assert torch.equal(torch.isnan(y0.float()), torch.isnan(y1.float()))
    assert torch.equal(torch.nan_to_num(y0.float()), torch.nan_to_num(y1.float()))
    if s0 is not None:
        torch.testing.assert_close(s1, s0, rtol=1e-6, atol=1e-3)  # the same stored values, summed in fixed point


def test_the_dispatcher_takes_pixel_rows_only_where_they_save_a_tile_round():
    """256 -> 256 @14x14 at batch 256: 225 padded positions per image make 15 units for one workgroup in 16 (two tiles of
    72 steps), 196 pixels make 12-13 (one tile): masked.  128 -> 128 @28x28: two tiles either way: padded (no masks)."""
    for shape, want in (((256, 256, 256, 14, 14), True), ((256, 128, 128, 28, 28), False)):
        B, Cin, Cout, H, W = shape
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, 9, Cin, device="cuda").to(torch.bfloat16)
        y = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
        d = _desc(B, Cin, Cout, H, W, Cin, Cout, 0, 0, flip=False)
        name = _run("1", d, x, w, y, None, None, None, None)
        assert ("masked" in name) == want, name


KSPLIT_SHAPES = [
    (256, 512, 512, 7, 7),    # CSPDarknet-53 stage 5 at batch 256: 6-7 units per workgroup, four filter tiles
    (128, 256, 256, 14, 14),  # stage 4 at the data-parallel per-GPU batch 128
    (192, 512, 256, 9, 11),   # odd map, rows not a multiple of 32
    (200, 256, 384, 8, 8),    # three filter tiles (do not divide 32), four chunks per group
]


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", KSPLIT_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_k_split_between_the_two_groups(shape, mode):
    """Round 5: where every workgroup's rows fit one group's tile, both groups take the same rows and half of the input
    channels each, and group 1's partial sums reach group 0 through LDS (span6 KSPLIT).  Against the kernel the dispatcher
    takes without it (VT_SPAN6_KSPLIT=0), on identical operands: another summation order -- two f32 partial sums added
    once -- so one bf16 rounding of the output apart at most (two with a residual), and against float64 on a sample."""
    B, Cin, Cout, H, W = shape
    flags = mode[1]
    torch.manual_seed(sum(shape) + 3)
    slices = shape[0] % 3 == 0
    ldx, ldy = (Cin + 32, Cout + 64) if slices else (Cin, Cout)
    xb = torch.randn(B, H, W, ldx, device="cuda").to(torch.bfloat16)
    x = xb[..., 16:16 + Cin] if slices else xb
    w = (torch.randn(Cout, 9, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if flags & N.VT_CONV_RESIDUAL else None
    scale = torch.rand(Cout, device="cuda") + 0.5 if flags & N.VT_CONV_AFFINE else None
    shift = torch.randn(Cout, device="cuda") if flags & N.VT_CONV_AFFINE else None
    d = _desc(B, Cin, Cout, H, W, ldx, ldy, Cout if res is not None else 0, flags, flip=mode[0] == "residual")
    outs = []
    try:
        for ks in (0, 1):
            N.set_knob("VT_SPAN6_KSPLIT", ks)
            yb = torch.full((B, H, W, ldy), float("nan"), device="cuda", dtype=torch.bfloat16)
            y = yb[..., 32:32 + Cout] if slices else yb
            st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
            name = _run("1", d, x, w, y, scale, shift, res, st)
            outs.append((yb, y, N.stats_decode(st) if st is not None else None, name))
    finally:
        N.set_knob("VT_SPAN6_KSPLIT", 1)
    (yb0, y0, s0, n0), (yb1, y1, s1, n1) = outs
    assert "ksplit" not in n0 and "ksplit" in n1, (n0, n1)
    assert torch.equal(torch.isnan(yb0.float()), torch.isnan(yb1.float()))  # nothing outside the slice was written
    a, b = y0.float(), y1.float()
    assert ((a - b).norm() / a.norm()).item() < 1e-3
    torch.testing.assert_close(b, a, rtol=2.0 ** -6, atol=2e-2)
    if s0 is not None:
        torch.testing.assert_close(s1, s0, rtol=2e-3, atol=0.5)
    # a float64 spot check of the plain convolution on one image (forward tap order only)
    if mode[0] in ("plain", "stats"):
        import torch.nn.functional as F
        xi = x[:1].permute(0, 3, 1, 2).double()
        wi = w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double()
        want = F.conv2d(xi, wi, None, 1, 1).permute(0, 2, 3, 1)
        assert ((y1[:1].double() - want).norm() / want.norm()).item() < 4e-3
