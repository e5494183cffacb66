"""Host-side logic that needs no GPU: drop-in surface (state_dict keys, attributes, factories),
the C-ABI export check, the launch-list compiler's structural invariants, and the
optimiser / schedule restatements."""
import ctypes
import json
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch
from torch import nn

from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox import engine as E
from vision_toolbox.components import ConvNormAct
from vision_toolbox.trainer import GROUP_BIAS, GROUP_NORM, GROUP_OTHER, TrainStep, param_groups, warmup_cosine_lr

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


# ---- drop-in surface -------------------------------------------------------------------------
def test_every_reference_factory_name_exists(manifest):
    for name in manifest:
        assert callable(getattr(backbones, name)), name
    for cls in ("Darknet", "DarknetYOLOv5", "VoVNet", "BaseBackbone"):
        assert hasattr(backbones, cls)


@pytest.mark.parametrize("name", ["darknet19", "darknet53", "cspdarknet53", "darknet_yolov5n", "darknet_yolov5x",
                                  "vovnet27_slim", "vovnet39", "vovnet19_slim_ese", "vovnet57_ese", "vovnet99_ese"])
def test_state_dict_keys_shapes_and_attributes_match_reference(name, manifest):
    ref = manifest[name]
    m = getattr(backbones, name)()
    got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    assert got == ref["keys"]
    assert sum(p.numel() for p in m.parameters()) == ref["num_parameters"]
    assert isinstance(m.out_channels_list, tuple) and all(isinstance(c, int) for c in m.out_channels_list)
    assert list(m.out_channels_list) == ref["out_channels_list"]
    assert isinstance(m.stride, int) and m.stride == ref["stride"]
    assert callable(m.get_feature_maps) and m.get_last_out_channels() == ref["out_channels_list"][-1]


def test_from_config_signatures():
    assert backbones.Darknet.from_config("cspdarknet53").out_channels_list == (64, 128, 256, 512, 1024)
    assert backbones.DarknetYOLOv5.from_config("x").out_channels_list == (80, 160, 320, 640, 1280)
    assert backbones.VoVNet.from_config(39, False, False).out_channels_list == (128, 256, 512, 768, 1024)
    with pytest.raises(KeyError):
        backbones.Darknet.from_config("nope")


def test_conv_norm_act_constructor_contract():
    for k, s in [(1, 1), (3, 1), (3, 2), (6, 2), (5, 1), (7, 2)]:
        u = ConvNormAct(4, 8, k, s)
        assert u.conv.padding == (-((s - k) // 2),) * 2  # ceil((k - s) / 2), components.py:31
        assert u.conv.bias is None and isinstance(u.norm, nn.BatchNorm2d) and isinstance(u.act, nn.ReLU)
    u = ConvNormAct(4, 8, norm="none", act="none")
    assert u.conv.bias is not None and isinstance(u.norm, nn.Identity)
    assert list(dict(ConvNormAct(4, 8).named_children())) == ["conv", "norm", "act"]
    # init: N(0, 2 / ((1 + 0.2^2) * fan_out)) for relu (components.py:45-46)
    torch.manual_seed(0)
    w = ConvNormAct(64, 256, 3).conv.weight
    assert w.std().item() == pytest.approx((2 / (1.04 * 256 * 9)) ** 0.5, rel=0.05)


def test_gpu_path_never_falls_back_to_the_eager_path(monkeypatch):
    """dispatch rule: CPU tensors -> the modules' own torch children; a GPU tensor must go to
    libvt_amd and fail loudly when the library is missing -- never silently through the eager path"""
    m = backbones.darknet19()
    assert m(torch.zeros(1, 3, 64, 64)).shape == (1, 1024, 2, 2)

    class _FakeCuda(torch.Tensor):
        is_cuda = True

    x = torch.zeros(1, 3, 64, 64).as_subclass(_FakeCuda)
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", ROOT / "nope.so")
    called = []
    monkeypatch.setattr(type(m), "_eager_maps", lambda self, x: called.append(1) or [])
    with pytest.raises(ImportError, match="no CPU/eager fallback"):
        m(x)
    assert not called


def test_parameter_groups_follow_classifier_py():
    model = nn.Sequential(backbones.cspdarknet53(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(1024, 1000))
    g = param_groups(model)
    counts = {k: sum(1 for v in g.values() if v == k) for k in (GROUP_NORM, GROUP_BIAS, GROUP_OTHER)}
    assert counts == {GROUP_NORM: 2 * 67, GROUP_BIAS: 1, GROUP_OTHER: 67 + 1}


# ---- C-ABI ---------------------------------------------------------------------------------------
def _declared_functions():
    text = (ROOT / "include" / "vt_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 30
    lib = N.lib()
    for n in names:
        assert n in N.SYMBOLS, f"{n} declared in vt_amd.h but not bound in _native.SYMBOLS"
        assert getattr(lib, n) is not None
    assert set(N.SYMBOLS) == set(names)
    assert lib.vt_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", str(N.LIB_PATH)], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (vt_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_ctypes_structs_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vt_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %d %d",'
                   "sizeof(vt_conv_desc), sizeof(vt_op), offsetof(vt_op, i), offsetof(vt_op, f),"
                   "offsetof(vt_conv_desc, dh), VT_OP_KIND_END, VT_STAT_REPLICAS);return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    vals = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()))
    assert vals[0] == ctypes.sizeof(N.ConvDesc)
    assert vals[1] == ctypes.sizeof(N.Op)
    assert vals[2] == N.Op.i.offset and vals[3] == N.Op.f.offset
    assert vals[4] == N.ConvDesc.dh.offset
    assert vals[5] == max(N.OP_NAMES) + 1
    assert vals[6] == N.VT_STAT_REPLICAS


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU/eager fallback"):
        N.lib()


# ---- launch-list compiler ----------------------------------------------------------------------------
def _dry_program(name, dtype, training, need_grad, size=64, batch=2):
    m = getattr(backbones, name)()
    m.train(training)
    r = m._vt_runner()
    r.store.ensure(torch.device("cpu"))
    return r.program(torch.zeros(batch, 3, size, size), dtype, True, need_grad)


def test_cspdarknet53_program_structure():
    p = _dry_program("cspdarknet53", N.VT_BF16, True, True)
    h = p.kind_histogram
    assert p.n_units == 67
    # the stem unit's backward is ONE streaming pass (BatchNorm reduction + filter-gradient correlations) and a
    # combine kernel: no dz, no bn_bwd_reduce / bn_bwd_apply / conv_wgrad for it (vt_stem_bwd.hip)
    assert h["stem_bwd_reduce"] == 1 and h["stem_bwd_combine"] == 1 and h["stem_bwd_s2"] == 1
    # 17 of the 38 1x1 units run as pointwise passes that recompute z (vt_pointwise.hip: the 32 / 64 / 128-channel ones
    # of stages 0-2), conv1 | conv2 of stages 0 and 1 as ONE two-group launch each: 15 launches per pass
    pw_units, pw_launches = 17, 15
    # (round 6: the apply passes finalize the BatchNorm coefficients for themselves -- pw_apply_fin / pw_bwd_fin)
    assert h["pw_stats"] == h["pw_apply_fin"] == h["pw_reduce"] == h["pw_bwd_fin"] == pw_launches
    # their filter gradient is formed inside pw_bwd up to 64 x 64; the 128-channel ones (stage 1 pair + out_conv, the 8
    # block units of stage 2: 11 filters) hand dz to the filter-gradient kernel
    assert h["conv_wgrad"] == 66 - pw_units + 11
    # round 6: the finalize step of the 49 units whose normalise / backward-apply passes are the streaming kernels runs
    # inside those passes (every workgroup finalizes its own channel group: vt_bn_finalize_apply / vt_bn_bwd_finalize_apply),
    # as do the pointwise units' apply passes; the stem keeps a finalize launch of its own
    assert h["bn_fin_apply"] == h["bn_bwd_fin_apply"] == 66 - pw_units and "bn_act_apply" not in h and "bn_bwd_apply" not in h
    assert h["bn_finalize"] == h["bn_bwd_finalize"] == 1  # (the stem)
    # (two other fused forms exist and are off by default, each measured no faster in the step -- VT_FUSE_BNRED=1 moves
    #  the reduction of DarknetBlock.conv1's backward into conv2's data-gradient launch, VT_BN_BWD_FUSED=1 makes the whole
    #  BatchNorm backward of a unit one launch; test_fused_backward_forms_change_the_program_as_documented)
    assert h["bn_bwd_reduce"] == 66 - pw_units
    # residual adds are folded into the normalise pass and torch.cat is elided: the ONLY
    # elementwise launches are one normalise pass (bn_fin_apply) per remaining unit; the only copies are the bf16 weight
    # mirror and the 3->8 channel stem filter pad
    # (the stem unit has none: its conv runs twice, statistics only and then with the normalise + ReLU epilogue, and
    # its pre-activation is never stored)
    assert h["copy2d"] == 2
    # forward convs + one data-gradient launch per conv; the 5 stride-2 convs take 4 parity-class launches, except the
    # two HBM-bound ones (32 -> 64 and 64 -> 128 channels), whose classes are the column blocks of one depth-to-space launch
    assert h["conv_igemm"] == (67 - pw_units) + 1 + (66 - 5 - pw_units) + 3 * 4 + 2
    assert "maxpool_fwd" not in h


def test_fused_backward_forms_change_the_program_as_documented(monkeypatch):
    monkeypatch.setenv("VT_FUSE_BNRED", "1")
    h = _dry_program("cspdarknet53", N.VT_BF16, True, True).kind_histogram
    # the 8 + 4 DarknetBlock.conv1 units of stages 3 and 4 (stages 0-2 are pointwise units under this file's VT_PW_MIN_MB=0)
    assert h["conv_dgrad_bnred"] == 12 and h["bn_bwd_reduce"] == 66 - 17 - 12 and h["bn_bwd_fin_apply"] == 66 - 17
    monkeypatch.setenv("VT_FUSE_BNRED", "0")
    monkeypatch.setenv("VT_BN_BWD_FUSED", "1")
    h = _dry_program("cspdarknet53", N.VT_BF16, True, True).kind_histogram
    assert h["bn_bwd_fused"] == 66 - 17 and "bn_bwd_fin_apply" not in h and "bn_bwd_reduce" not in h
    assert h["bn_bwd_finalize"] == 1  # (the stem; the pointwise units finalize inside pw_bwd_fin)
    monkeypatch.setenv("VT_BN_BWD_FUSED", "0")
    monkeypatch.setenv("VT_BN_FIN_APPLY", "0")  # finalize launches of their own (rounds 1-5)
    h = _dry_program("cspdarknet53", N.VT_BF16, True, True).kind_histogram
    assert h["bn_act_apply"] == h["bn_bwd_apply"] == h["bn_bwd_reduce"] == 66 - 17 and h["bn_finalize"] == h["bn_bwd_finalize"] == 67
    assert "bn_fin_apply" not in h and "bn_bwd_fin_apply" not in h


def test_inference_program_is_fully_fused():
    p = _dry_program("cspdarknet53", N.VT_BF16, False, False)
    h = p.kind_histogram
    # one launch per unit and nothing else: a conv with the affine + ReLU epilogue, or -- for the 1x1 units the pointwise
    # kernels cover (17 of them in 15 launches, as in training; large inputs only outside this test's VT_PW_MIN_MB=0) --
    # the pointwise apply pass with the running-statistics coefficients
    assert h["conv_igemm"] == 67 - 17 and h["pw_apply"] == 15 and "pw_stats" not in h
    assert "bn_act_apply" not in h and p.n_bwd == 0
    p = _dry_program("vovnet39", N.VT_F32, False, False)
    assert p.kind_histogram["conv_igemm"] == 39 and p.kind_histogram["maxpool_fwd"] == 4
    # OSA concat elided, f32 needs no mirror: the one copy is the stem filter's 3 -> 4 channel pad
    assert p.kind_histogram["copy2d"] == 1 and "bn_act_apply" not in p.kind_histogram


def test_vovnet_ese_program_compiles_with_gradients():
    p = _dry_program("vovnet19_slim_ese", N.VT_F32, True, True)
    h = p.kind_histogram
    assert h["ese_fwd"] == 4 and h["ese_bwd"] == 4 and h["maxpool_bwd"] == 4
    assert h["avgpool_fwd"] == 4 and h["colsum"] == 4


def test_unsupported_variants_raise_not_fallback():
    # (round 5: ConvNormAct's own activation choices all have kernels -- their codes, components.py:37-44 ...)
    assert [ConvNormAct(8, 8, act=a)._vt_relu() for a in ("none", "relu", "leaky_relu", "swish", "silu", "gelu")] == \
        [0, 1, 2, 3, 3, 4]
    # ... an activation that is NOT one of them raises, it does not fall back
    u = ConvNormAct(8, 8, act="gelu")
    u.act = torch.nn.GELU(approximate="tanh")
    b = E.Builder(E.ParamStore(u), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    x = b.act(1, 4, 4, 8)
    with pytest.raises(NotImplementedError, match="activation"):
        u._vt_emit(b, x)
    # the generic activations emit the unfused passes (conv, coefficients, normalise + activation) also in eval mode
    u = ConvNormAct(8, 8, act="silu").eval()
    b = E.Builder(E.ParamStore(u), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    u._vt_emit(b, b.act(1, 4, 4, 8))
    kinds = [op.kind & 0xFFFF for op in b.fwd]
    assert N.OP_BN_ACT_APPLY in kinds and N.OP_BN_EVAL_COEFFS in kinds
    act_op = [op for op in b.fwd if (op.kind & 0xFFFF) == N.OP_BN_ACT_APPLY][0]
    assert act_op.i[4] == 3  # the activation code travels in the op
    # groups: one unit per group over channel slices (two conv launches, each 4 -> 4 channels of pixel stride 8) ...
    g = ConvNormAct(8, 8, groups=2).eval()
    b = E.Builder(E.ParamStore(g), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    g._vt_emit(b, b.act(1, 4, 4, 8))
    convs = [op for op in b.fwd if (op.kind & 0xFFFF) == N.OP_CONV_IGEMM]
    assert len(convs) == 2
    descs = [N.ConvDesc.from_buffer_copy(bytes(op.i)[: ctypes.sizeof(N.ConvDesc)]) for op in convs]
    assert [(d.Cin, d.ldx, d.Cout, d.ldy, d.ldw) for d in descs] == [(4, 8, 4, 8, 36)] * 2
    assert convs[1].ptr[1].offset - convs[0].ptr[1].offset == 4 * 36 * 4  # the second group's filter rows
    # ... depthwise (groups == channels, round 6): one launch of the depthwise kernel, no per-group units ...
    g = ConvNormAct(8, 8, groups=8).eval()
    b = E.Builder(E.ParamStore(g), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    g._vt_emit(b, b.act(1, 4, 4, 8))
    kinds = [op.kind & 0xFFFF for op in b.fwd]
    assert kinds.count(N.OP_DWCONV_FWD) == 1 and N.OP_CONV_IGEMM not in kinds
    # ... but not a group below one 16-byte chunk that is not depthwise: raises, no fallback
    g = ConvNormAct(8, 8, groups=4)
    b = E.Builder(E.ParamStore(g), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    with pytest.raises(NotImplementedError, match="16-byte chunk"):
        g._vt_emit(b, b.act(1, 4, 4, 8))
    # dilation: the taps move apart, the padding stays ceil((k - s) / 2) (components.py:31), so the map shrinks
    dl = ConvNormAct(8, 8, dilation=2).eval()
    b = E.Builder(E.ParamStore(dl), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    y = dl._vt_emit(b, b.act(1, 9, 7, 8))
    assert (y.H, y.W) == (7, 5)
    d = N.ConvDesc.from_buffer_copy(bytes([op for op in b.fwd if (op.kind & 0xFFFF) == N.OP_CONV_IGEMM][0].i)[: ctypes.sizeof(N.ConvDesc)])
    assert [(d.dh[i], d.dw[i]) for i in range(9)] == [(2 * r, 2 * t) for r in range(3) for t in range(3)]
    # norm="none": biased conv, then the activation through the unit-scale normalise pass
    nn_ = ConvNormAct(8, 8, norm="none", act="leaky_relu").eval()
    b = E.Builder(E.ParamStore(nn_), N.VT_F32, False, False)
    b.store.ensure(torch.device("cpu"))
    nn_._vt_emit(b, b.act(1, 4, 4, 8))
    kinds = [op.kind & 0xFFFF for op in b.fwd]
    assert kinds.count(N.OP_CONV_IGEMM) == 1 and N.OP_BN_ACT_APPLY in kinds and N.OP_BN_FINALIZE not in kinds
    act_op = [op for op in b.fwd if (op.kind & 0xFFFF) == N.OP_BN_ACT_APPLY][0]
    assert act_op.i[4] == 2 and act_op.ptr[1].base == -1 and act_op.ptr[2].base == -1  # no scale / shift


def test_param_store_keeps_identity_names_and_values():
    m = backbones.darknet_yolov5n()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ids = {k: id(p) for k, p in m.named_parameters()}
    st = E.ParamStore(m)
    st.ensure(torch.device("cpu"))
    after = m.state_dict()
    assert list(after) == list(before)
    for k in before:
        assert torch.equal(after[k], before[k]), k
    assert ids == {k: id(p) for k, p in m.named_parameters()}
    w = m.stem.conv.weight
    assert w.shape == (16, 3, 6, 6) and w.stride() == (108, 1, 18, 3)  # OIHW view of [O][kh][kw][I]
    assert not st.stale(torch.device("cpu"))
    m.stem.conv.weight.data = m.stem.conv.weight.data.clone()
    assert st.stale(torch.device("cpu"))


def test_train_step_plan_covers_all_parameters():
    ts = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True)
    segs = ts.segments
    assert segs[0][0] == 0 and segs[-1][1] == ts.store.pflat.numel()
    assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    assert [s[2] for s in segs] == [0.0, 0.0, 2e-5]  # norm, bias, everything else
    assert ts.prog.kind_histogram["xent"] == 1 and ts.n_opt == 3
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.step()


def test_warmup_cosine_matches_torch_schedulers():
    from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.5)
    cos = CosineAnnealingLR(opt, T_max=95, eta_min=0.0)
    sched = SequentialLR(opt, [LinearLR(opt, start_factor=0.01, total_iters=5), cos], milestones=[5])
    for epoch in range(100):
        assert opt.param_groups[0]["lr"] == pytest.approx(warmup_cosine_lr(epoch, 100, 0.5), rel=1e-6, abs=1e-9)
        opt.step()
        sched.step()


def test_statistics_buffer_fixed_point_roundtrip():
    """include/vt_amd.h: a statistics buffer is int64[VT_STAT_REPLICAS][2][C][2], value = hi*2^12 + lo/2^33; the
    Python helpers the tests use to fill / read one follow the kernels' split (vt_common.h vt_stat_add): good for
    any sign and magnitude to 2^-34 absolute, and additive across replicas."""
    C_ = 7
    buf = torch.zeros(N.VT_STAT_REPLICAS, 2, C_, 2, dtype=torch.int64)
    assert buf.numel() * 8 == N.stat_floats(C_) * 4
    v = torch.tensor([0.0, 1e-6, -1e-6, 3.25, -4095.75, 4096.0, 1.2345e7], dtype=torch.float32)
    w = torch.tensor([5e-9, -7.5, 123456.7, -0.03125, 4095.999, -8192.5, -1.2345e7], dtype=torch.float32)
    N.stats_encode(buf, 0, v, replica=3)
    N.stats_encode(buf, 0, w, replica=N.VT_STAT_REPLICAS - 1)
    N.stats_encode(buf, 1, w, replica=0)
    got = N.stats_decode(buf)
    # resolution 2^-33 per contribution (values with finer bits, 1e-6 or 5e-9, round to it): 2^-34 error each
    assert (got[0] - (v.double() + w.double())).abs().max().item() <= 2 * 2.0 ** -34
    assert (got[1] - w.double()).abs().max().item() <= 2.0 ** -34
    assert got[0][3].item() == 3.25 - 0.03125 and got[0][5].item() == 4096.0 - 8192.5  # dyadic values: exact
    hi = buf[3, 0, :, 0]
    assert hi.tolist() == [0, 0, 0, 0, 0, 1, 3013]  # trunc(v / 4096): the hi limb stays zero below 4096


def test_padded_stem_filter_is_read_from_the_bf16_mirror():
    """under exchange='sharded' only the bf16 mirror of a conv filter is refreshed on the ranks that do
    not own its slice, so no forward op may read a conv filter from the f32 master buffer in bf16 mode -- the RGB stem's
    padding copy (3 -> 8 channels) did.  It now takes the mirror (same rounding of the same master value)."""
    import torch

    from vision_toolbox import _native as N
    from vision_toolbox import backbones
    from vision_toolbox import engine as E
    from vision_toolbox.trainer import TrainStep

    for factory in (backbones.cspdarknet53, backbones.darknet_yolov5n, backbones.vovnet19_slim_ese):
        ts = TrainStep(factory(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True)
        st = ts.store
        conv_w = [(o, o + p.numel()) for o, p in zip(st.offsets, st.params) if p.dim() == 4]
        copies = 0
        for i in range(ts.prog.n_fwd):
            op = ts.prog.fwd_ops[i]
            for k in range(N.VT_OP_MAX_PTR):
                if op.ptr[k].base == E.PARAMS:  # an f32 read of the master buffer: BatchNorm / bias parameters only
                    off = op.ptr[k].offset // 4
                    assert not any(a <= off < b for a, b in conv_w), (factory.__name__, i, op.kind & 0xFFFF)
            if (op.kind & 0xFFFF) == N.OP_COPY2D and op.ptr[0].base == E.MIRROR:
                copies += 1
        assert copies == 1, factory.__name__
