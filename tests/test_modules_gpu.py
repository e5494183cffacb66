"""Module-level parity on the GPU: the shipped vision_toolbox modules (HIP path) against
the golden vectors produced by the unmodified reference (tests/golden, tools/gen_golden.py)
and against the CPU oracle.  Also a restatement of the reference's own
tests/test_backbones.py:39-78 (attributes / forward / get_feature_maps / jit.trace) on cuda.

Stated tolerances: f32 kernels 1e-3 relative on logits (north_star), 2e-4 relative L2 on
unit/block tensors.  bf16 kernels store every activation AND every gradient in bf16
(2^-8 relative rounding each): 3e-2 relative L2 on forward tensors; on the toy-sized
(tens of samples per channel) unit/block cases the BatchNorm backward cancels most of the
incoming gradient, which amplifies that rounding, so gradients get 0.25 -- the f32 path,
which shares every line of code except the MFMA opcode and the rounding, carries the
strict check.
"""
import copy
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import filler
from oracle import torch_ref as R
from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox.backbones import Darknet, DarknetYOLOv5, VoVNet
from vision_toolbox.backbones.darknet import CSPDarknetStage, DarknetBlock, DarknetStage
from vision_toolbox.backbones.vovnet import OSABlock
from vision_toolbox.components import ConvNormAct

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

F32_TOL, BF16_TOL = 2e-4, 3e-2


@pytest.fixture(autouse=True)
def _hip_path_ran():
    N.lib()
    before = N.launch_count()
    yield
    torch.cuda.synchronize()
    assert N.launch_count() > before, "no libvt_amd launch happened: the HIP path did not run"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _run_case(m: nn.Module, tag: str, g, x_shape, dtype):
    """train fwd+bwd and eval fwd of a shipped module vs the golden arrays of `tag`."""
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    filler.fill_module(m, tag + ".")
    m = m.cuda()
    m.compute_dtype = dtype
    for sub in m.modules():
        if hasattr(sub, "compute_dtype"):
            sub.compute_dtype = dtype
    x = filler.tensor(tag + ".x", x_shape).cuda().requires_grad_(True)
    m.train()
    y = m(x)
    assert y.shape == tuple(g[tag + ".y"].shape)
    gy = filler.tensor(tag + ".gy", y.shape).cuda()
    y.backward(gy.to(y.dtype))
    assert rel_err(y.detach().float().cpu(), _t(g[tag + ".y"])) < tol, "forward"
    gtol = 4 * tol if dtype == torch.float32 else 0.25
    assert rel_err(x.grad.float().cpu(), _t(g[tag + ".dx"])) < gtol, "dx"
    for k, p in m.named_parameters():
        ref = _t(g[f"{tag}.grad.{k}"])
        assert p.grad is not None, k
        err = (p.grad.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-3 * (ref.numel() ** 0.5))
        assert err < gtol, f"grad {k}: {err}"
    for k, b in m.named_buffers():
        ref = _t(g[f"{tag}.buf.{k}"])
        if b.dtype == torch.int64:
            assert int(b) == int(ref), k
        else:
            np.testing.assert_allclose(b.cpu().numpy(), ref.numpy(), rtol=10 * tol, atol=10 * tol, err_msg=k)
    filler.fill_module(m, tag + ".")
    m.eval()
    with torch.no_grad():
        ye = m(x.detach())
    assert rel_err(ye.float().cpu(), _t(g[tag + ".y_eval"])) < tol, "eval forward"


UNIT_CASES = [(16, 32, 1, 1, 8), (16, 16, 3, 1, 9), (8, 24, 3, 2, 10), (8, 16, 6, 2, 12), (3, 16, 3, 1, 10),
              (3, 16, 6, 2, 12), (3, 16, 3, 2, 11)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", UNIT_CASES, ids=lambda c: "_".join(map(str, c)))
def test_conv_norm_act_unit_vs_reference(case, dtype, golden_dir):
    cin, cout, k, s, hw = case
    g = np.load(golden_dir / "units.npz")
    tag = f"cna_{cin}_{cout}_k{k}s{s}_{hw}"
    _run_case(ConvNormAct(cin, cout, k, s), tag, g, (2, cin, hw, hw), dtype)


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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("tag", sorted(BLOCK_CASES))
def test_block_vs_reference(tag, dtype, golden_dir):
    g = np.load(golden_dir / "blocks.npz")
    f, shape = BLOCK_CASES[tag]
    _run_case(f(), tag, g, shape, dtype)


MODELS = ["darknet19", "cspdarknet53", "darknet53", "darknet_yolov5n", "darknet_yolov5x", "vovnet39",
          "vovnet19_slim_ese", "vovnet27_slim"]


def _classifier(name, dtype):
    """model assembly of classifier.py:58-64 around the shipped backbone."""
    bb = getattr(backbones, name)()
    model = nn.Sequential(bb, nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(bb.get_last_out_channels(), 16))
    filler.fill_module(model, name + ".")
    bb.compute_dtype = dtype
    return model.cuda()


@pytest.mark.parametrize("name", MODELS)
def test_model_train_step_f32_vs_reference(name, golden_dir):
    gm = np.load(golden_dir / "models.npz")
    model = _classifier(name, torch.float32)
    x, y = filler.images(4, 64).cuda(), filler.labels(4, 16).cuda()
    model.train()
    logits = model(x)
    loss = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss.backward()
    ref_logits = _t(gm[f"{name}.train.logits"])
    # north_star: forward logits within 1e-3 rel-tol of the CPU reference
    assert rel_err(logits.detach().cpu(), ref_logits) < 1e-3
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref_logits.numpy(), rtol=1e-3,
                               atol=1e-3 * ref_logits.abs().max().item())
    assert loss.item() == pytest.approx(float(gm[f"{name}.train.loss"]), rel=1e-4)
    # Train-mode gradients at this toy size (4 images @64px: 16 samples per channel in the last
    # stage) are ill-conditioned: the forward already amplifies f32 rounding ~100x (logits agree
    # to ~1e-4), and backward runs the same chain again, so the stem-side gradients of the deep
    # nets differ by up to ~1e-2 between ANY two f32 implementations.  The tight backward check
    # is test_model_eval_mode_gradients_f32_vs_reference below.
    keys = list(gm[f"{name}.train.grad_keys"])
    norms = gm[f"{name}.train.grad_norms"]
    params = dict(model.named_parameters())
    got = np.array([params[k].grad.double().norm().item() for k in keys])
    rel = np.abs(got - norms) / np.maximum(norms, 1e-4 * norms.max())
    assert np.median(rel) < 5e-3 and rel.max() < 0.1, (np.median(rel), rel.max())
    for k in (keys[0], keys[len(keys) // 2], "3.weight", "3.bias"):
        ref = _t(gm[f"{name}.train.grad.{k}"])
        assert rel_err(params[k].grad.cpu(), ref) < 0.1, k
    sd = model.state_dict()
    first_bn = next(k for k in sd if k.endswith("running_mean"))
    np.testing.assert_allclose(sd[first_bn].cpu().numpy(), gm[f"{name}.train.first_running_mean"], rtol=1e-4, atol=1e-5)
    last_bn = [k for k in sd if k.endswith("running_var")][-1]
    np.testing.assert_allclose(sd[last_bn].cpu().numpy(), gm[f"{name}.train.last_running_var"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", MODELS)
def test_model_eval_mode_gradients_f32_vs_reference(name, golden_dir):
    """BatchNorm in eval mode (running statistics are constants) with gradients enabled: the
    whole backward wiring -- data/filter gradients of every conv incl. the stride-2 parity
    classes, residual/concat gradient routing, pooling, ESE -- in a well-conditioned setting."""
    gm = np.load(golden_dir / "models.npz")
    model = _classifier(name, torch.float32).eval()
    x, y = filler.images(4, 64).cuda(), filler.labels(4, 16).cuda()
    loss = F.cross_entropy(model(x), y, label_smoothing=0.1)
    loss.backward()
    assert loss.item() == pytest.approx(float(gm[f"{name}.evalgrad.loss"]), rel=1e-4)
    keys = list(gm[f"{name}.train.grad_keys"])
    params = dict(model.named_parameters())
    got = np.array([params[k].grad.double().norm().item() for k in keys])
    ref = gm[f"{name}.evalgrad.grad_norms"]
    # A wiring error shows up as O(1) errors on MANY parameters.  What remains legitimately is
    # the ReLU boundary: an element whose pre-activation is within f32 rounding of 0 can land
    # on the other side of the mask than in ATen (z*scale+shift as one fma here vs ATen's
    # (z-mean)*invstd*gamma+beta); ONE such element (measured: 1 of 16384 in yolov5n stage 0,
    # tools/debug_stage0.py) moves every gradient upstream of it by ~1e-2.  Hence: the typical
    # (median) parameter must match tightly, the worst one loosely.
    rel = np.abs(got - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert np.median(rel) < 1e-4 and rel.max() < 5e-2, (np.median(rel), rel.max())
    assert rel_err(params["3.weight"].grad.cpu(), _t(gm[f"{name}.evalgrad.grad.3.weight"])) < 1e-4
    for k in (keys[0], keys[1], keys[len(keys) // 2]):
        assert rel_err(params[k].grad.cpu(), _t(gm[f"{name}.evalgrad.grad.{k}"])) < 5e-2, k


@pytest.mark.parametrize("name", MODELS)
def test_model_eval_feature_maps_f32_vs_reference(name, golden_dir):
    gm = np.load(golden_dir / "models.npz")
    model = _classifier(name, torch.float32).eval()
    x = filler.images(4, 64).cuda()
    with torch.no_grad():
        maps = model[0].get_feature_maps(x)
        logits = model(x)
    assert len(maps) == len(model[0].out_channels_list)
    for i, (mp, c) in enumerate(zip(maps, model[0].out_channels_list)):
        assert mp.shape[1] == c
        flat = mp.contiguous().reshape(-1).float().cpu()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        ref = _t(gm[f"{name}.eval.map{i}.samples"])
        scale = float(gm[f"{name}.eval.map{i}.summary"][1]) + 1e-6
        assert (flat[idx] - ref).abs().max().item() < 1e-3 * max(scale, ref.abs().max().item())
        assert mp.double().norm().item() == pytest.approx(float(gm[f"{name}.eval.map{i}.summary"][2]), rel=1e-4)
    assert rel_err(logits.cpu(), _t(gm[f"{name}.eval.logits"])) < 1e-3


@pytest.mark.parametrize("pw_min_mb", ["0", "80"], ids=["pointwise_units", "production_threshold"])
@pytest.mark.parametrize("name", ["cspdarknet53", "vovnet39", "darknet19"])
def test_model_bf16_tracks_reference(name, pw_min_mb, golden_dir, monkeypatch):
    """bf16 kernels: every activation is stored in bf16, so only a loose bound is meaningful.  Both settings of the
    pointwise threshold (read when the launch lists are built): every covered 1x1 unit on vt_pointwise.hip, and the
    production default, under which these toy tensors take the unfused conv + BatchNorm kernels."""
    monkeypatch.setenv("VT_PW_MIN_MB", pw_min_mb)
    gm = np.load(golden_dir / "models.npz")
    model = _classifier(name, torch.bfloat16)
    x, y = filler.images(4, 64).cuda(), filler.labels(4, 16).cuda()
    model.eval()
    with torch.no_grad():
        logits = model[3](model[2](model[1](model[0](x).float())))
    ref = _t(gm[f"{name}.eval.logits"])
    assert rel_err(logits.cpu(), ref) < 5e-2
    model.train()
    logits = model[3](model[2](model[1](model[0](x).float())))
    loss = F.cross_entropy(logits, y, label_smoothing=0.1)
    loss.backward()
    assert loss.item() == pytest.approx(float(gm[f"{name}.train.loss"]), rel=5e-2)
    keys = list(gm[f"{name}.train.grad_keys"])
    norms = gm[f"{name}.train.grad_norms"]
    params = dict(model.named_parameters())
    got = np.array([params[k].grad.double().norm().item() for k in keys])
    big = norms > 1e-3 * norms.max()
    # At this toy size bf16 is ill-conditioned: one bf16 rounding of a first-stage activation (8.8e-5 of that map's
    # norm) becomes 12 % of the last VoVNet map, because BatchNorm there normalises 2x2 maps (16 values per channel at
    # batch 4).  While the statistics were f32 atomics the result fell into two run-to-run classes (median 0.027 / loss
    # 3.8059 in one run out of four, 0.102 / 3.8454 otherwise; golden loss 3.8066); with the fixed-point statistics
    # (vt_common.h) every run gives 0.102.  CSPDarknet-53: 0.035, Darknet-19: 0.010.  The tight bf16 checks are the
    # batch-256 unit tests against a bf16-storage-emulating float64 reference (test_fullsize_gpu.py).
    assert np.median(np.abs(got[big] / norms[big] - 1)) < 0.15


@pytest.mark.parametrize("name", ["vovnet39", "cspdarknet53"])
def test_training_pass_is_bit_reproducible_up_to_the_filter_gradients(name):
    """BatchNorm statistics and backward sums are 64-bit fixed-point integer atomics (vt_common.h): every activation,
    the loss, the data gradients and the BatchNorm parameter gradients must be BIT-IDENTICAL from run to run (with f32
    atomics this toy VoVNet-39 fell into two classes 12 % apart at the last feature map).  Only the conv filter
    gradients still go through f32 atomics (split over pixel ranges): equal to ~1e-6."""
    runs = []
    for _ in range(4):
        model = _classifier(name, torch.bfloat16)
        model.train()
        x = filler.images(4, 64).cuda().requires_grad_(True)
        y = filler.labels(4, 16).cuda()
        maps = model[0].get_feature_maps(x)
        logits = model[3](model[2](model[1](maps[-1].float())))
        loss = F.cross_entropy(logits, y, label_smoothing=0.1)
        loss.backward()
        params = dict(model.named_parameters())
        runs.append(([m.detach().clone() for m in maps], loss.detach().clone(), x.grad.clone(),
                     {k: v.grad.clone() for k, v in params.items()}))
    maps0, loss0, dx0, g0 = runs[0]
    for maps, loss, dx, g in runs[1:]:
        for a, b in zip(maps, maps0):
            assert torch.equal(a, b)
        assert torch.equal(loss, loss0) and torch.equal(dx, dx0)
        for k in g0:
            if ".norm." in k:
                assert torch.equal(g[k], g0[k]), k
            else:
                assert ((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item() < 1e-5, k


def test_config1_darknet19_224_forward(golden_dir):
    """BASELINE.json configs[0] (Darknet-19, 1x3x224x224): HIP forward vs the reference's CPU forward."""
    gm = np.load(golden_dir / "models.npz")
    model = _classifier("darknet19", torch.float32).eval()
    x = filler.images(1, 224, seed=224).cuda()
    with torch.no_grad():
        f = model[0](x)
        logits = model(x)
    assert tuple(f.shape) == (1, 1024, 7, 7)
    flat = f.contiguous().reshape(-1).float().cpu()
    idx = torch.linspace(0, flat.numel() - 1, 256).long()
    ref = _t(gm["darknet19.cfg1.last.samples"])
    assert (flat[idx] - ref).abs().max().item() < 1e-3 * ref.abs().max().item()
    assert rel_err(logits.cpu(), _t(gm["darknet19.cfg1.logits"])) < 1e-3


# ---- restatement of the reference's own backbone tests (tests/test_backbones.py:39-78) on cuda
FACTORIES = [
    lambda: Darknet.from_config("darknet19"),
    lambda: Darknet.from_config("cspdarknet53"),
    lambda: DarknetYOLOv5.from_config("n"),
    lambda: DarknetYOLOv5.from_config("l"),
    lambda: VoVNet.from_config(27, True, False),
    lambda: VoVNet.from_config(39, False, False),
    lambda: VoVNet.from_config(19, True, True),
    lambda: VoVNet.from_config(57, False, True),
]
FACTORY_IDS = ["darknet19", "cspdarknet53", "yolov5n", "yolov5l", "vovnet27_slim", "vovnet39", "vovnet19_slim_ese",
               "vovnet57_ese"]


@pytest.fixture
def inputs():
    return torch.rand(1, 3, 224, 224, device="cuda")


@pytest.mark.parametrize("factory", FACTORIES, ids=FACTORY_IDS)
class TestBackbone:
    def test_forward(self, factory, inputs):
        m = factory().cuda()
        out = m(inputs)
        assert isinstance(out, torch.Tensor) and out.dim() == 4

    def test_get_feature_maps(self, factory, inputs):
        m = factory().cuda()
        outs = m.get_feature_maps(inputs)
        assert isinstance(outs, list) and len(outs) == len(m.out_channels_list)
        for o, c in zip(outs, m.out_channels_list):
            assert isinstance(o, torch.Tensor) and o.dim() == 4 and o.shape[1] == c
            assert o.requires_grad  # ordinary autograd-tracked tensors

    def test_jit_trace(self, factory, inputs):
        m = factory().cuda().eval()
        traced = torch.jit.trace(m, inputs, check_trace=False)
        torch.testing.assert_close(traced(inputs), m(inputs), rtol=1e-4, atol=1e-4)


def test_traced_graph_holds_the_registered_operator_and_survives_save_load(tmp_path):
    """SURVEY 8b / VERDICT r1 missing 6: the module API dispatches to `vision_toolbox_amd::backbone` (torch.library),
    so a trace is a real operator node -- serialisable, unlike an autograd.Function's PythonOp."""
    m = backbones.darknet19().cuda().eval()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    traced = torch.jit.trace(m, x, check_trace=False)
    assert "vision_toolbox_amd::backbone" in str(traced.graph)
    path = str(tmp_path / "darknet19_traced.pt")
    traced.save(path)
    loaded = torch.jit.load(path)
    torch.testing.assert_close(loaded(x), m(x), rtol=1e-5, atol=1e-5)


def test_torch_compile_traces_through_the_operator():
    """the fake-tensor implementation gives shapes / strides / dtypes, so torch.compile (aot_eager: dynamo + AOT
    autograd tracing with fake tensors, no inductor code generation) runs the module and matches eager; reference
    tests/test_backbones.py:80-86."""
    m = backbones.vovnet19_slim_ese().cuda().eval()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    want = [t.clone() for t in m.get_feature_maps(x)]
    compiled = torch.compile(m.get_feature_maps, backend="aot_eager")
    got = compiled(x)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_operator_autograd_matches_module_autograd():
    """gradients through the registered autograd formula (input and every parameter) after a second, interleaved
    forward: each graph owns its own run state."""
    m = backbones.darknet_yolov5n().cuda().train()
    filler.fill_module(m, "opg.")
    x1 = torch.rand(2, 3, 64, 64, device="cuda", requires_grad=True)
    x2 = torch.rand(2, 3, 64, 64, device="cuda", requires_grad=True)
    y1 = m(x1)
    y2 = m(x2)  # a second forward before the first backward
    (y1.float().square().mean()).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    dx1 = x1.grad.clone()
    for p in m.parameters():
        p.grad = None
    (y2.float().square().mean()).backward()
    assert x2.grad is not None and torch.isfinite(x2.grad).all()
    # the same forward alone gives the same gradients (batch statistics make the two forwards independent)
    for p in m.parameters():
        p.grad = None
    x1b = x1.detach().clone().requires_grad_(True)
    m(x1b).float().square().mean().backward()
    # (f32 atomics in the statistics arrive in a different order from run to run: compare in relative L2)
    assert rel_err(x1b.grad, dx1) < 1e-3
    for k, p in m.named_parameters():
        assert rel_err(p.grad, g1[k]) < 5e-3, k


def test_oracle_and_hip_agree_on_fresh_random_init():
    """default init (kaiming fan_out, components.py:45-46) -> copy weights to the oracle."""
    torch.manual_seed(0)
    m = backbones.cspdarknet53()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = filler.images(2, 96, seed=7)
    with torch.no_grad():
        ref = R.feature_maps("cspdarknet53", sd, x, False)
    m = m.cuda().eval()
    with torch.no_grad():
        got = m.get_feature_maps(x.cuda())
    for a, b in zip(got, ref):
        assert rel_err(a.float().cpu(), b) < 1e-3


def test_load_state_dict_after_first_run():
    m = backbones.darknet19().cuda().eval()
    x = filler.images(1, 64).cuda()
    with torch.no_grad():
        y0 = m(x).clone()
    new = {k: filler.fill_tensor("reload." + k, v.cpu()) for k, v in m.state_dict().items()}
    m.load_state_dict(new)  # checkpoints in the reference's format (base.py:23-25)
    with torch.no_grad():
        y1 = m(x)
        ref = R.feature_maps("darknet19", {k: v.clone() for k, v in new.items()}, x.cpu(), False)[-1]
    assert not torch.allclose(y0, y1)
    assert rel_err(y1.float().cpu(), ref) < 1e-3
    with pytest.raises(RuntimeError):  # parameters live on the GPU: a CPU tensor fails in torch, as in the reference
        m(x.cpu())


def test_accumulates_into_existing_grad_and_works_with_torch_optim():
    m = ConvNormAct(8, 16, 3, 1).cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    x = filler.tensor("acc.x", (2, 8, 6, 6)).cuda()
    m(x).sum().backward()
    g1 = m.conv.weight.grad.clone()
    m(x).sum().backward()
    torch.testing.assert_close(m.conv.weight.grad, 2 * g1, rtol=1e-4, atol=1e-5)
    w0 = m.conv.weight.detach().clone()
    opt.step()
    assert not torch.equal(w0, m.conv.weight.detach())
    y = m(x)  # runs with the updated weights without any manual sync
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("name,shape", [("darknet19", (2, 3, 75, 91)), ("cspdarknet53", (3, 3, 160, 96)),
                                        ("vovnet27_slim", (2, 3, 75, 91)), ("darknet_yolov5s", (5, 3, 64, 200)),
                                        ("vovnet39", (2, 3, 130, 130)), ("cspdarknet53", (1, 3, 33, 47))],
                         ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_odd_input_shapes_match_the_oracle(name, shape, training):
    """The reference takes any image size (backbones/darknet.py:83-87, vovnet.py:100-104: stride-2 convs and
    MaxPool2d(3, 2, 1) on odd maps, non-square images, batch sizes that are no multiple of anything): every feature map
    of the f32 path against the CPU oracle on inputs that are not 224 x 224 (tools/odd_shapes_check.py runs the full
    grid of 5 models x 5 shapes x 2 modes)."""
    m = getattr(backbones, name)()
    filler.fill_module(m, f"odd.{name}.")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = filler.tensor(f"odd{shape}", shape)
    ref = R.feature_maps(name, sd, x, training)
    m = m.cuda().train(training)
    with torch.no_grad():
        maps = m.get_feature_maps(x.cuda())
    assert [tuple(t.shape) for t in maps] == [tuple(t.shape) for t in ref]
    for got, want in zip(maps, ref):
        # (a train-mode map of four values per channel -- 33 x 47 at stride 32 -- amplifies f32 rounding: 2e-3 there)
        assert rel_err(got.cpu(), want) < (2e-3 if min(want.shape[2:]) <= 2 and training else 2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("act", ["leaky_relu", "swish", "silu", "gelu", "none"])
@pytest.mark.parametrize("k,s", [(3, 1), (1, 1), (3, 2)])
def test_conv_norm_act_other_activations_forward_backward(act, k, s, dtype):
    """ConvNormAct's non-default activations (reference components.py:37-44: leaky_relu(0.2), swish / silu, gelu, none) on the
    GPU path, train and eval mode, against the module's own torch children on the CPU in float64 (the CPU path of the SAME
    class is the reference's arithmetic: nn.Conv2d -> nn.BatchNorm2d -> the activation)."""
    from vision_toolbox.components import ConvNormAct

    def rel_err(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

    torch.manual_seed(7)
    m = ConvNormAct(16, 24, k, s, act=act)
    with torch.no_grad():
        m.norm.weight.uniform_(0.5, 1.5)
        m.norm.bias.uniform_(-0.3, 0.3)
        m.norm.running_mean.uniform_(-0.2, 0.2)
        m.norm.running_var.uniform_(0.5, 1.5)
    ref = copy.deepcopy(m).double()
    x = torch.randn(4, 16, 13, 11)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    for training in (True, False):
        m.train(training), ref.train(training)
        xr = x.double().requires_grad_(True)
        yr = ref.act(ref.norm(ref.conv(xr)))
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3)).double()
        ref.zero_grad()
        yr.backward(gy)
        dev = m.cuda()
        dev.compute_dtype = dtype
        xd = x.cuda().requires_grad_(True)
        before = N.launch_count()
        yd = dev(xd)
        dev.zero_grad()
        yd.backward(gy.float().cuda())
        torch.cuda.synchronize()
        assert N.launch_count() > before
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert rel_err(yd.float().cpu(), yr.detach()) < tol, (act, training, "y")
        assert rel_err(xd.grad.float().cpu(), xr.grad) < (1e-4 if dtype == torch.float32 else 4e-2), (act, training, "dx")
        assert rel_err(dev.conv.weight.grad.float().cpu(), ref.conv.weight.grad) < (1e-4 if dtype == torch.float32 else 4e-2)
        assert rel_err(dev.norm.weight.grad.float().cpu(), ref.norm.weight.grad) < (1e-4 if dtype == torch.float32 else 4e-2)
        m = dev.cpu()


def _variant_check(m, ref, x, dtype, fwd_ref, params, bf16_slack=1.0):
    """one train and one eval pass of HipModule `m` on the GPU against the float64 `ref` (same parameters) on the CPU"""
    f32 = dtype == torch.float32
    for training in (True, False):
        # (bf16: the caller gives `ref` the bf16-ROUNDED filter the kernels read -- an unrounded one flips the sign of a
        #  few pre-activations near zero, and one flipped mask element moves a gradient that is a sum of ~500 random-sign
        #  terms by percents; with the same operands what remains is the rounding of the stored activations: 2e-3 to 1.7e-2, 3e-2 allowed.
        #  In train mode BatchNorm backward subtracts the gradient's per-channel projections (most of it, at ~500 samples per
        #  channel) and so amplifies that rounding (module docstring): 1.7-3.9e-2 measured, 8e-2 allowed; f32 is the strict run.
        gtol = 1e-4 if f32 else (8e-2 if training else 3e-2) * bf16_slack
        m.train(training), ref.train(training)
        xr = x.double().requires_grad_(True)
        yr = fwd_ref(ref, xr)
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3)).double()
        ref.zero_grad()
        yr.backward(gy)
        dev = m.cuda()
        dev.compute_dtype = dtype
        xd = x.cuda().requires_grad_(True)
        before = N.launch_count()
        yd = dev(xd)
        dev.zero_grad()
        yd.backward(gy.float().cuda())
        torch.cuda.synchronize()
        assert N.launch_count() > before
        assert tuple(yd.shape) == tuple(yr.shape)
        assert rel_err(yd.float().cpu(), yr.detach()) < (2e-5 if f32 else 1e-2), (training, "y")
        assert rel_err(xd.grad.float().cpu(), xr.grad) < gtol, (training, "dx")
        got = dict(dev.named_parameters())
        for name, p in ref.named_parameters():
            if name in params:
                assert got[name].grad is not None, name
                assert rel_err(got[name].grad.float().cpu(), p.grad) < gtol, (training, name)
        if training and hasattr(ref, "norm") and isinstance(ref.norm, nn.BatchNorm2d):
            assert rel_err(dev.norm.running_var.cpu(), ref.norm.running_var) < 1e-3
            assert int(dev.norm.num_batches_tracked) == int(ref.norm.num_batches_tracked)
        m = dev.cpu()


VARIANTS = [  # Cin, Cout, k, s, kwargs
    (16, 32, 3, 1, dict(groups=2)),
    (32, 32, 3, 2, dict(groups=4)),
    (32, 16, 1, 1, dict(groups=2)),
    (16, 24, 3, 1, dict(dilation=2)),
    (16, 24, 3, 2, dict(dilation=2)),
    (8, 16, 3, 1, dict(dilation=3)),
    (16, 24, 3, 1, dict(norm="none", act="relu")),
    (16, 24, 1, 1, dict(norm="none", act="leaky_relu")),
    (16, 24, 3, 2, dict(norm="none", act="gelu")),
    (16, 24, 3, 1, dict(norm="none", act="none")),
    (32, 32, 3, 1, dict(groups=2, dilation=2, norm="none", act="silu")),
    (32, 32, 3, 2, dict(groups=2, dilation=2, act="leaky_relu")),
    # round 6: depthwise (groups = in_channels = out_channels; vt_dwconv.hip)
    (16, 16, 3, 1, dict(groups=16)),
    (24, 24, 3, 2, dict(groups=24)),
    (16, 16, 5, 1, dict(groups=16, dilation=2, act="silu")),
    (32, 32, 3, 2, dict(groups=32, norm="none", act="relu")),
    (16, 16, 1, 1, dict(groups=16, norm="none", act="none")),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", VARIANTS, ids=lambda c: f"{c[0]}-{c[1]}k{c[2]}s{c[3]}" + "".join(f"_{k}{v}" for k, v in c[4].items()))
def test_conv_norm_act_groups_dilation_and_no_norm(case, dtype):
    """The remaining constructor arguments of ConvNormAct (reference components.py:13-36): `groups` (one unit per group
    over channel slices), `dilation` (taps apart, padding unchanged: the map shrinks) and norm="none" (biased conv, then
    the activation) -- forward, input gradient and every parameter gradient, train and eval, against the module's own
    torch children in float64 on the CPU."""
    Cin, Cout, k, s, kw = case
    torch.manual_seed(11)
    m = ConvNormAct(Cin, Cout, k, s, **kw)
    with torch.no_grad():
        if isinstance(m.norm, nn.BatchNorm2d):
            m.norm.weight.uniform_(0.5, 1.5)
            m.norm.bias.uniform_(-0.3, 0.3)
            m.norm.running_mean.uniform_(-0.2, 0.2)
            m.norm.running_var.uniform_(0.5, 1.5)
        else:
            m.conv.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(m).double()
    x = torch.randn(4, Cin, 13, 11)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        with torch.no_grad():
            ref.conv.weight.copy_(m.conv.weight.bfloat16().double())
    names = {"conv.weight", "conv.bias", "norm.weight", "norm.bias"}
    # (depthwise, bf16: a filter-gradient entry is a sum over the 572 pixels of ONE channel -- nothing averages a flipped ReLU
    #  mask element or a rounded dz out over input channels; measured 3.3e-2 in eval mode where dense units stay under
    #  1.7e-2.  The f32 run of the same case carries the 1e-4 bound.)
    depthwise = kw.get("groups", 1) == Cin == Cout
    _variant_check(m, ref, x, dtype, lambda r, xr: r.act(r.norm(r.conv(xr))), names, bf16_slack=2.0 if depthwise else 1.0)


def test_grouped_unit_inside_a_chain_with_a_shortcut():
    """A grouped unit between ordinary ones, with a shortcut around it: its input's gradient is written slice by slice by
    the groups and ALSO receives a full-width identity contribution, and its output's gradient arrives full width (from the
    next unit's data gradient) and is read slice by slice -- the cases the gradient bookkeeping has to split."""
    from vision_toolbox.components import HipModule

    class Chain(HipModule):
        def __init__(self):
            super().__init__()
            self.a = ConvNormAct(8, 32, 3, 1)
            self.g = ConvNormAct(32, 32, 3, 1, groups=4)
            self.c = ConvNormAct(32, 16, 1, 1)
            self.out_channels_list = (16,)
            self.stride = 1

        def _vt_emit_maps(self, b, x):
            y = self.a._vt_emit(b, x)
            y2 = self.g._vt_emit(b, y, residual=y)
            return [self.c._vt_emit(b, y2)]

        def _eager_maps(self, x):
            y = self.a(x)
            return [self.c(self.g(y) + y)]

    torch.manual_seed(5)
    m = Chain()
    ref = copy.deepcopy(m).double()
    x = torch.randn(3, 8, 10, 9)
    _variant_check(m, ref, x, torch.float32, lambda r, xr: r._eager_maps(xr)[-1],
                   {n for n, _ in m.named_parameters()})
