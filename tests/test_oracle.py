"""Pin the CPU oracle (oracle/torch_ref.py) against the golden vectors that
tools/gen_golden.py produced from the unmodified reference.  CPU only."""
import json
import math

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import torch_ref as R

README_PARAMS_M = {  # reference README.md:128-135,176-183
    "darknet19": 19.82, "darknet53": 40.58, "cspdarknet53": 26.24, "darknet_yolov5n": 0.88,
    "darknet_yolov5s": 3.51, "darknet_yolov5m": 10.69, "darknet_yolov5l": 23.96, "darknet_yolov5x": 45.18,
    "vovnet27_slim": 3.01, "vovnet39": 21.58, "vovnet57": 35.62, "vovnet19_slim_ese": 2.68,
    "vovnet19_ese": 10.18, "vovnet39_ese": 25.18, "vovnet57_ese": 41.45, "vovnet99_ese": 69.52,
}  # fmt: skip


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


@pytest.mark.parametrize("name", sorted(R.FACTORIES))
def test_spec_matches_reference_state_dict(name, manifest):
    ref = manifest[name]
    got = [[k, list(s)] for k, s in R.spec(name).items()]
    assert got == ref["keys"]
    assert R.num_parameters(name) == ref["num_parameters"]
    assert list(R.out_channels_list(name)) == ref["out_channels_list"]
    assert round(ref["num_parameters"] / 1e6, 2) == pytest.approx(README_PARAMS_M[name], abs=0.011)


def _filled(name, prefix, spec=None):
    sd = R.empty_state_dict(name) if spec is None else spec
    return {k: filler.fill_tensor(prefix + k, v) for k, v in sd.items()}


def _clf_sd(name, num_classes):
    sd = {}
    for k, shape in R.classifier_spec(name, num_classes).items():
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = filler.fill_tensor(f"{name}.{k}", torch.zeros(shape, dtype=dt))
    return sd


MODELS = ["darknet19", "cspdarknet53", "darknet53", "darknet_yolov5n", "darknet_yolov5x", "vovnet39",
          "vovnet19_slim_ese", "vovnet27_slim"]


@pytest.fixture(scope="module")
def gm(golden_dir):
    return np.load(golden_dir / "models.npz")


@pytest.mark.parametrize("name", MODELS)
def test_oracle_train_step_matches_reference(name, gm):
    sd = _clf_sd(name, 16)
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x, y = filler.images(4, 64), filler.labels(4, 16)
    loss, logits = R.classifier_loss(name, sd, x, y, 0.1, training=True)
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), gm[f"{name}.train.logits"], rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(float(gm[f"{name}.train.loss"]), rel=1e-5)
    keys = list(gm[f"{name}.train.grad_keys"])
    norms = gm[f"{name}.train.grad_norms"]
    got = np.array([sd[k].grad.double().norm().item() for k in keys])
    np.testing.assert_allclose(got, norms, rtol=2e-3, atol=1e-7)
    first_bn = next(k for k in sd if k.endswith("running_mean"))
    np.testing.assert_allclose(sd[first_bn].numpy(), gm[f"{name}.train.first_running_mean"], rtol=1e-5, atol=1e-6)
    last_bn = [k for k in sd if k.endswith("running_var")][-1]
    np.testing.assert_allclose(sd[last_bn].numpy(), gm[f"{name}.train.last_running_var"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", MODELS)
def test_oracle_eval_mode_gradients_match_reference(name, gm):
    sd = _clf_sd(name, 16)
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    loss, _ = R.classifier_loss(name, sd, filler.images(4, 64), filler.labels(4, 16), 0.1, training=False)
    loss.backward()
    assert loss.item() == pytest.approx(float(gm[f"{name}.evalgrad.loss"]), rel=1e-5)
    keys = list(gm[f"{name}.train.grad_keys"])
    got = np.array([sd[k].grad.double().norm().item() for k in keys])
    np.testing.assert_allclose(got, gm[f"{name}.evalgrad.grad_norms"], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("name", MODELS)
def test_oracle_eval_feature_maps_match_reference(name, gm):
    sd = _clf_sd(name, 16)
    x = filler.images(4, 64)
    with torch.no_grad():
        maps = R.feature_maps(name, sd, x, False, prefix="0.")
        logits = R.classifier_logits(name, sd, x, False)
    assert len(maps) == len(R.out_channels_list(name))
    for i, m in enumerate(maps):
        flat = m.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        np.testing.assert_allclose(flat[idx].numpy(), gm[f"{name}.eval.map{i}.samples"], rtol=1e-4, atol=1e-5)
        s = gm[f"{name}.eval.map{i}.summary"]
        assert m.double().norm().item() == pytest.approx(s[2], rel=1e-5)
    np.testing.assert_allclose(logits.numpy(), gm[f"{name}.eval.logits"], rtol=1e-4, atol=1e-5)


def test_oracle_config1_darknet19_224(gm):
    """BASELINE.json configs[0]: Darknet-19 forward on CPU, 1x3x224x224."""
    sd = _clf_sd("darknet19", 16)
    x = filler.images(1, 224, seed=224)
    with torch.no_grad():
        f = R.feature_maps("darknet19", sd, x, False, prefix="0.")[-1]
        logits = R.classifier_logits("darknet19", sd, x, False)
    assert tuple(f.shape) == (1, 1024, 7, 7)
    flat = f.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 256).long()
    np.testing.assert_allclose(flat[idx].numpy(), gm["darknet19.cfg1.last.samples"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), gm["darknet19.cfg1.logits"], rtol=1e-4, atol=1e-5)


def test_padding_rule():
    """padding = ceil((k - s) / 2) (components.py:31) -> output sizes of the four (k, s) on the path."""
    for k, s, h in [(1, 1, 9), (3, 1, 9), (3, 2, 9), (3, 2, 10), (6, 2, 12), (6, 2, 13)]:
        sd = {"conv.weight": torch.zeros(4, 4, k, k), "norm.weight": torch.ones(4), "norm.bias": torch.zeros(4),
              "norm.running_mean": torch.zeros(4), "norm.running_var": torch.ones(4),
              "norm.num_batches_tracked": torch.zeros((), dtype=torch.int64)}
        y = R.cna(sd, "", torch.zeros(1, 4, h, h), s, False)
        pad = math.ceil((k - s) / 2)
        assert y.shape[-1] == (h + 2 * pad - k) // s + 1


def test_sgd_restatement_matches_torch_optim():
    torch.manual_seed(0)
    p = {"a.weight": torch.randn(5, 3), "a.norm.weight": torch.randn(7), "a.bias": torch.randn(5)}
    q = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    opt = torch.optim.SGD(
        [{"params": [q["a.norm.weight"]], "weight_decay": 0.0}, {"params": [q["a.bias"]], "weight_decay": 0.0},
         {"params": [q["a.weight"]], "weight_decay": 2e-5}], lr=0.05, momentum=0.9)
    mom = {}
    for step in range(3):
        grads = {k: torch.randn_like(v) for k, v in p.items()}
        for k in q:
            q[k].grad = grads[k].clone()
        opt.step()
        R.sgd_step(p, grads, mom, 0.05, 0.9, lambda k: R.weight_decay_group(k, 2e-5, 0.0, 0.0))
    for k in p:
        torch.testing.assert_close(p[k], q[k].data, rtol=1e-6, atol=1e-7)


def test_oracle_config5_yolov5x_640(gm):
    """BASELINE.json configs[4]: Darknet-YOLOv5x get_feature_maps() @640px (two images; eval mode)."""
    sd = {k: filler.fill_tensor("darknet_yolov5x.cfg5." + k, v) for k, v in R.empty_state_dict("darknet_yolov5x").items()}
    x = filler.images(2, 640, seed=640)
    with torch.no_grad():
        maps = R.feature_maps("darknet_yolov5x", sd, x, False)
    assert [tuple(m.shape) for m in maps] == [tuple(gm[f"darknet_yolov5x.cfg5.map{i}.shape"]) for i in range(5)]
    for i, m in enumerate(maps):
        for b in range(2):
            flat = m[b].reshape(-1)
            idx = torch.linspace(0, flat.numel() - 1, 512).long()
            ref = gm[f"darknet_yolov5x.cfg5.map{i}.img{b}.samples"]
            np.testing.assert_allclose(flat[idx].numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
