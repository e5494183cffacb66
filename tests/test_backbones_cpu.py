"""The dispatch rule of SURVEY 8(b): the SAME shipped module classes on CPU tensors.

Part 1 is the reference's own tests/test_backbones.py:19-78 (CPU input `torch.rand(1, 3, 224, 224)`,
same factory list, same five checks per factory) run against this package.  Part 2 pins the CPU path
numerically against the golden vectors generated from the unmodified reference (tools/gen_golden.py)
-- the eager path is built from the modules' own nn.Conv2d / nn.BatchNorm2d / nn.ReLU children and must
match ATen to f32 rounding -- and runs BASELINE.json configs[0] literally:
`backbones.darknet19()(x)` on CPU, 1x3x224x224.  Nothing here touches libvt_amd or oracle/torch_ref.
"""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from oracle import filler  # deterministic weight / input filler only (no arithmetic)
from vision_toolbox import _native as N
from vision_toolbox import backbones, necks
from vision_toolbox.backbones import Darknet, DarknetYOLOv5, VoVNet
from vision_toolbox.backbones.darknet import CSPDarknetStage, DarknetBlock, DarknetStage
from vision_toolbox.backbones.vovnet import OSABlock
from vision_toolbox.components import ConvNormAct


@pytest.fixture
def inputs():
    return torch.rand(1, 3, 224, 224)


factory_list = [
    *[partial(Darknet.from_config, x) for x in ("darknet19", "cspdarknet53")],
    *[partial(DarknetYOLOv5.from_config, x) for x in ("n", "l")],
    *[
        partial(VoVNet.from_config, x, y, z)
        for x, y, z in ((27, True, False), (39, False, False), (19, True, True), (57, False, True))
    ],
]


@pytest.fixture(autouse=True)
def _no_native_launch():
    """the CPU path must not go anywhere near the HIP library"""
    before = N.launch_count()
    yield
    assert N.launch_count() == before


@pytest.mark.parametrize("factory", factory_list)
class TestBackbone:
    def test_attributes(self, factory):
        m = factory()

        assert hasattr(m, "out_channels_list")
        assert isinstance(m.out_channels_list, tuple)
        for c in m.out_channels_list:
            assert isinstance(c, int)

        assert hasattr(m, "stride")
        assert isinstance(m.stride, int)

        assert hasattr(m, "get_feature_maps")
        assert callable(m.get_feature_maps)

    def test_forward(self, factory, inputs):
        m = factory()
        outputs = m(inputs)

        assert isinstance(outputs, Tensor)
        assert len(outputs.shape) == 4

    def test_get_feature_maps(self, factory, inputs):
        m = factory()
        outputs = m.get_feature_maps(inputs)

        assert isinstance(outputs, list)
        assert len(outputs) == len(m.out_channels_list)
        for out, out_c in zip(outputs, m.out_channels_list):
            assert isinstance(out, Tensor)
            assert len(out.shape) == 4
            assert out.shape[1] == out_c

    def test_pretrained(self, factory, monkeypatch):
        """the reference downloads the checkpoint (base.py:23-25); offline, torch.hub is replaced by a
        loader that serves a reference-format state dict for the requested URL"""
        served = {}

        def fake_hub(url, *a, **k):
            sd = {k2: filler.fill_tensor("ckpt." + k2, v) for k2, v in factory().state_dict().items()}
            served[url] = sd
            return sd

        monkeypatch.setattr(torch.hub, "load_state_dict_from_url", fake_hub)
        m = factory(pretrained=True)
        (url, sd), = served.items()
        assert url.startswith("https://github.com/gau-nernst/vision-toolbox/releases/download/") and url.endswith(".pth")
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd[k]), k

    def test_jit_trace(self, factory, inputs):
        m = factory()
        traced = torch.jit.trace(m, inputs)
        # on CPU the trace is plain aten ops, so it also serialises
        m.eval()
        traced = torch.jit.trace(m, inputs)
        torch.testing.assert_close(traced(inputs), m(inputs))


# ---- numerical pin of the CPU path against the unmodified reference ------------------------------
def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _run_case(m: nn.Module, tag: str, g, x_shape):
    filler.fill_module(m, tag + ".")
    x = filler.tensor(tag + ".x", x_shape).requires_grad_(True)
    m.train()
    y = m(x)
    y.backward(filler.tensor(tag + ".gy", y.shape))
    assert _rel(y.detach(), _t(g[tag + ".y"])) < 1e-6
    assert _rel(x.grad, _t(g[tag + ".dx"])) < 1e-5
    for k, p in m.named_parameters():
        ref = _t(g[f"{tag}.grad.{k}"])
        assert (p.grad - ref).norm() / ref.norm().clamp_min(1e-3 * ref.numel() ** 0.5) < 1e-4, k
    for k, b in m.named_buffers():
        np.testing.assert_allclose(b.numpy(), g[f"{tag}.buf.{k}"], rtol=1e-5, atol=1e-6, err_msg=k)
    filler.fill_module(m, tag + ".")
    m.eval()
    with torch.no_grad():
        assert _rel(m(x.detach()), _t(g[tag + ".y_eval"])) < 1e-6


UNIT_CASES = [(16, 32, 1, 1, 8), (16, 16, 3, 1, 9), (8, 24, 3, 2, 10), (8, 16, 6, 2, 12), (3, 16, 3, 1, 10),
              (3, 16, 6, 2, 12), (3, 16, 3, 2, 11)]


@pytest.mark.parametrize("case", UNIT_CASES, ids=lambda c: "_".join(map(str, c)))
def test_conv_norm_act_cpu_vs_reference(case, golden_dir):
    cin, cout, k, s, hw = case
    _run_case(ConvNormAct(cin, cout, k, s), f"cna_{cin}_{cout}_k{k}s{s}_{hw}", np.load(golden_dir / "units.npz"),
              (2, cin, hw, hw))


BLOCK_CASES = {
    "darknet_block_16": (lambda: DarknetBlock(16), (2, 16, 6, 6)),
    "darknet_block_e1_16": (lambda: DarknetBlock(16, expansion=1), (2, 16, 6, 6)),
    "darknet_stage_2_8_16": (lambda: DarknetStage(2, 8, 16), (2, 8, 10, 10)),
    "csp_stage_1_8_16": (lambda: CSPDarknetStage(1, 8, 16), (2, 8, 10, 10)),
    "csp_stage_2_16_32": (lambda: CSPDarknetStage(2, 16, 32), (2, 16, 9, 9)),
    "osa_16_8_3_32": (lambda: OSABlock(16, 8, 3, 32, ese=False), (2, 16, 7, 7)),
    "osa_16_8_3_16_res": (lambda: OSABlock(16, 8, 3, 16, ese=False), (2, 16, 7, 7)),
    "osa_16_8_3_16_res_ese": (lambda: OSABlock(16, 8, 3, 16, ese=True), (2, 16, 7, 7)),
    "osa_16_8_2_24_ese": (lambda: OSABlock(16, 8, 2, 24, ese=True), (2, 16, 6, 6)),
}


@pytest.mark.parametrize("tag", sorted(BLOCK_CASES))
def test_block_cpu_vs_reference(tag, golden_dir):
    f, shape = BLOCK_CASES[tag]
    _run_case(f(), tag, np.load(golden_dir / "blocks.npz"), shape)


def _classifier(name):
    bb = getattr(backbones, name)()
    model = nn.Sequential(bb, nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(bb.get_last_out_channels(), 16))
    filler.fill_module(model, name + ".")
    return model


@pytest.mark.parametrize("name", ["darknet19", "cspdarknet53", "darknet_yolov5n", "darknet_yolov5x", "vovnet39",
                                  "vovnet19_slim_ese"])
def test_model_cpu_vs_reference(name, golden_dir):
    """classifier.py:58-64 assembly around the shipped backbone, run on CPU: train-mode logits, loss and
    per-parameter gradient norms, eval-mode feature maps, against the unmodified reference"""
    gm = np.load(golden_dir / "models.npz")
    model = _classifier(name)
    x, y = filler.images(4, 64), filler.labels(4, 16)
    model.train()
    logits = model(x)
    loss = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), gm[f"{name}.train.logits"], rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(float(gm[f"{name}.train.loss"]), rel=1e-5)
    params = dict(model.named_parameters())
    keys = list(gm[f"{name}.train.grad_keys"])
    got = np.array([params[k].grad.double().norm().item() for k in keys])
    np.testing.assert_allclose(got, gm[f"{name}.train.grad_norms"], rtol=2e-3, atol=1e-7)
    filler.fill_module(model, name + ".")
    model.eval()
    with torch.no_grad():
        maps = model[0].get_feature_maps(x)
    for i, mp in enumerate(maps):
        flat = mp.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        np.testing.assert_allclose(flat[idx].numpy(), gm[f"{name}.eval.map{i}.samples"], rtol=1e-4, atol=1e-5)


def test_config1_darknet19_forward_on_cpu(golden_dir):
    """BASELINE.json configs[0], literally: Darknet-19 forward on CPU, 1x3x224x224, via backbones.darknet19()"""
    gm = np.load(golden_dir / "models.npz")
    model = _classifier("darknet19").eval()
    x = filler.images(1, 224, seed=224)
    with torch.no_grad():
        f = model[0](x)
        logits = model(x)
    assert tuple(f.shape) == (1, 1024, 7, 7) and f.device.type == "cpu"
    flat = f.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 256).long()
    np.testing.assert_allclose(flat[idx].numpy(), gm["darknet19.cfg1.last.samples"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), gm["darknet19.cfg1.logits"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_necks_cpu_vs_reference(golden_dir, mode):
    """FPN / PAN on CPU tensors (same dispatch rule) against the unmodified reference's fixtures"""
    from test_necks import CASES, _case, _inputs, _loss, _make  # same case table as the oracle / GPU tests

    g = np.load(golden_dir / "necks.npz")
    for name in CASES:  # (round 6: fuse_fn="concat" and interpolation_mode="bilinear" cases too)
        kind, ins, outc, td, sizes, B, fuse, interp = _case(name)
        m = _make(necks, name)
        filler.fill_module(m, name + ".")
        m.train(mode == "train")
        xs = _inputs(name, ins, sizes, B)
        ys = m(xs)
        _loss(name, ys).backward()
        for i, y in enumerate(ys):
            assert _rel(y.detach(), _t(g[f"{name}/{mode}/y{i}"])) < 1e-6, (name, i)
        for i, x in enumerate(xs):
            assert _rel(x.grad, _t(g[f"{name}/{mode}/dx{i}"])) < 1e-5, (name, i)
        for k, p in m.named_parameters():
            assert _rel(p.grad, _t(g[f"{name}/{mode}/grad/{k}"])) < 1e-5, (name, k)


def test_frozen_batchnorm_and_requires_grad_reach_the_compiled_program():
    """a `bn.eval()` inside a training model, or a parameter frozen / unfrozen after the first
    call, must change the compiled launch lists (they are part of the program cache key)"""
    m = backbones.darknet_yolov5n().train()
    r = m._vt_runner()
    r.store.ensure(torch.device("cpu"))
    x = torch.zeros(2, 3, 64, 64)
    p0 = r.program(x, N.VT_F32, True, True)
    n_units = p0.n_units
    # (round 6: the finalize step of a training-mode BatchNorm runs inside the normalise pass, vt_bn_finalize_apply)
    assert p0.kind_histogram["bn_fin_apply"] == n_units and p0.kind_histogram["conv_wgrad"] == n_units
    m.stem.norm.eval()  # frozen BatchNorm: running statistics, which must not be updated
    p1 = r.program(x, N.VT_F32, True, True)
    assert p1 is not p0
    assert p1.kind_histogram["bn_fin_apply"] == n_units - 1 and p1.kind_histogram["bn_eval_coeffs"] == 1
    m.stem.conv.weight.requires_grad_(False)
    p2 = r.program(x, N.VT_F32, True, True)
    assert p2 is not p1 and p2.kind_histogram["conv_wgrad"] == n_units - 1
    m.stem.conv.weight.requires_grad_(True)
    m.stem.norm.train()
    assert r.program(x, N.VT_F32, True, True) is p0
