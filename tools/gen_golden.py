"""Generate tests/golden/* by running the UNMODIFIED reference hot path on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes are
data (inputs are regenerated from oracle/filler.py, outputs are stored) and are committed.
The reference cannot be imported as shipped here because `torchvision` is absent
(vision_toolbox/components.py:7); the shim below stubs that one off-path symbol and
bypasses the two eager package __init__s -- no reference source is modified or copied.

    python tools/gen_golden.py
"""
from __future__ import annotations

import importlib
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import filler  # noqa: E402

REF = Path("/root/reference")
GOLDEN = ROOT / "tests" / "golden"


_REF_MODULES: dict = {}  # the reference's modules, kept out of sys.modules between uses


class reference_namespace:
    """Inside the block `vision_toolbox` IS the unmodified reference package (behind the torchvision shim); outside
    it the name is whatever it was before (this repo's package, in a pytest run: tests/test_checkpoint.py imports this
    module in the middle of one, and the two packages share their import name)."""

    @staticmethod
    def _mine(k):
        return k == "vision_toolbox" or k.startswith("vision_toolbox.")

    def __enter__(self):
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        ops.DeformConv2d = type("DeformConv2d", (nn.Module,), {})
        tv.ops = ops
        sys.modules.setdefault("torchvision", tv)
        sys.modules.setdefault("torchvision.ops", ops)
        self.saved = {k: sys.modules.pop(k) for k in list(sys.modules) if self._mine(k)}
        if not _REF_MODULES:
            pkg = types.ModuleType("vision_toolbox")
            pkg.__path__ = [str(REF / "vision_toolbox")]
            sub = types.ModuleType("vision_toolbox.backbones")
            sub.__path__ = [str(REF / "vision_toolbox" / "backbones")]
            _REF_MODULES["vision_toolbox"] = pkg
            _REF_MODULES["vision_toolbox.backbones"] = sub
        sys.modules.update(_REF_MODULES)
        return self

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if self._mine(k)]:
            _REF_MODULES[k] = sys.modules.pop(k)
        sys.modules.update(self.saved)
        return False


def ref_import(name: str):
    """import a module of the unmodified reference, e.g. ref_import("vision_toolbox.necks")"""
    with reference_namespace():
        return importlib.import_module(name)


def import_reference():
    with reference_namespace():
        comp = importlib.import_module("vision_toolbox.components")
        dk = importlib.import_module("vision_toolbox.backbones.darknet")
        vv = importlib.import_module("vision_toolbox.backbones.vovnet")
    return comp, dk, vv


comp, dk, vv = import_reference()

FACTORIES = {
    "darknet19": lambda: dk.Darknet.from_config("darknet19"),
    "darknet53": lambda: dk.Darknet.from_config("darknet53"),
    "cspdarknet53": lambda: dk.Darknet.from_config("cspdarknet53"),
    **{f"darknet_yolov5{v}": (lambda v=v: dk.DarknetYOLOv5.from_config(v)) for v in "nsmlx"},
    "vovnet27_slim": lambda: vv.VoVNet.from_config(27, True, False),
    "vovnet39": lambda: vv.VoVNet.from_config(39, False, False),
    "vovnet57": lambda: vv.VoVNet.from_config(57, False, False),
    "vovnet19_slim_ese": lambda: vv.VoVNet.from_config(19, True, True),
    "vovnet19_ese": lambda: vv.VoVNet.from_config(19, False, True),
    "vovnet39_ese": lambda: vv.VoVNet.from_config(39, False, True),
    "vovnet57_ese": lambda: vv.VoVNet.from_config(57, False, True),
    "vovnet99_ese": lambda: vv.VoVNet.from_config(99, False, True),
}


def np_(t):
    return t.detach().cpu().numpy().copy()  # copy: module tensors are overwritten in place later


def samples(t: torch.Tensor, n: int = 64):
    flat = t.detach().reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, n).long()
    return np_(flat[idx])


def summary(t: torch.Tensor):
    t = t.detach().double()
    return np.array([t.mean().item(), t.std().item(), t.norm().item()], dtype=np.float64)


# ---------------------------------------------------------------------------------------
def gen_manifest():
    man = {}
    for name, f in FACTORIES.items():
        m = f()
        sd = m.state_dict()
        man[name] = {
            "keys": [[k, list(v.shape)] for k, v in sd.items()],
            "num_parameters": sum(p.numel() for p in m.parameters()),
            "out_channels_list": list(m.out_channels_list),
            "stride": int(m.stride),
        }
        with torch.no_grad():
            m.eval()
            maps = m.get_feature_maps(torch.zeros(1, 3, 64, 64))
        man[name]["map_shapes_64"] = [list(o.shape) for o in maps]
    (GOLDEN / "manifest.json").write_text(json.dumps(man, indent=0))
    print("manifest:", {k: v["num_parameters"] for k, v in man.items()})


UNIT_CASES = [  # (cin, cout, k, s, hw)
    (16, 32, 1, 1, 8),
    (16, 16, 3, 1, 9),
    (8, 24, 3, 2, 10),
    (8, 16, 6, 2, 12),
    (3, 16, 3, 1, 10),
    (3, 16, 6, 2, 12),
    (3, 16, 3, 2, 11),
]


def run_module(m, x, tag, out: dict, prefix: str):
    """train-mode fwd+bwd and eval-mode fwd of a reference module on filler weights."""
    filler.fill_module(m, prefix)
    m.train()
    xx = x.clone().requires_grad_(True)
    y = m(xx)
    gy = filler.tensor(tag + ".gy", y.shape)
    y.backward(gy)
    out[tag + ".y"] = np_(y)
    out[tag + ".dx"] = np_(xx.grad)
    for k, p in m.named_parameters():
        out[f"{tag}.grad.{k}"] = np_(p.grad)
    for k, b in m.named_buffers():
        out[f"{tag}.buf.{k}"] = np_(b)
    filler.fill_module(m, prefix)
    m.eval()
    with torch.no_grad():
        out[tag + ".y_eval"] = np_(m(x))


def gen_units():
    out = {}
    for cin, cout, k, s, hw in UNIT_CASES:
        tag = f"cna_{cin}_{cout}_k{k}s{s}_{hw}"
        m = comp.ConvNormAct(cin, cout, k, s)
        x = filler.tensor(tag + ".x", (2, cin, hw, hw))
        run_module(m, x, tag, out, tag + ".")
    np.savez_compressed(GOLDEN / "units.npz", **out)
    print("units:", len(out), "arrays")


BLOCK_CASES = {
    "darknet_block_16": (lambda: dk.DarknetBlock(16), (2, 16, 6, 6)),
    "darknet_block_e1_16": (lambda: dk.DarknetBlock(16, expansion=1), (2, 16, 6, 6)),
    "darknet_stage_2_8_16": (lambda: dk.DarknetStage(2, 8, 16), (2, 8, 10, 10)),
    "csp_stage_1_8_16": (lambda: dk.CSPDarknetStage(1, 8, 16), (2, 8, 10, 10)),
    "csp_stage_2_16_32": (lambda: dk.CSPDarknetStage(2, 16, 32), (2, 16, 9, 9)),
    "osa_16_8_3_32": (lambda: vv.OSABlock(16, 8, 3, 32, ese=False), (2, 16, 7, 7)),
    "osa_16_8_3_16_res": (lambda: vv.OSABlock(16, 8, 3, 16, ese=False), (2, 16, 7, 7)),
    "osa_16_8_3_16_res_ese": (lambda: vv.OSABlock(16, 8, 3, 16, ese=True), (2, 16, 7, 7)),
    "osa_16_8_2_24_ese": (lambda: vv.OSABlock(16, 8, 2, 24, ese=True), (2, 16, 6, 6)),
}


def gen_blocks():
    out = {}
    for tag, (f, shape) in BLOCK_CASES.items():
        x = filler.tensor(tag + ".x", shape)
        run_module(f(), x, tag, out, tag + ".")
    np.savez_compressed(GOLDEN / "blocks.npz", **out)
    print("blocks:", len(out), "arrays")


MODEL_CASES = ["darknet19", "cspdarknet53", "darknet53", "darknet_yolov5n", "darknet_yolov5x", "vovnet39",
               "vovnet19_slim_ese", "vovnet27_slim"]
NUM_CLASSES, LABEL_SMOOTHING = 16, 0.1


def make_classifier(name):
    """the model assembly of classifier.py:58-64 (restated: Lightning is not importable here)."""
    bb = FACTORIES[name]()
    return nn.Sequential(bb, nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(bb.out_channels_list[-1], NUM_CLASSES))


def gen_models():
    out = {}
    for name in MODEL_CASES:
        model = make_classifier(name)
        filler.fill_module(model, name + ".")
        x = filler.images(4, 64)
        y = filler.labels(4, NUM_CLASSES)
        # train step: forward, loss (classifier.py:91-92), backward
        model.train()
        maps = model[0].get_feature_maps(x)
        for i, mp in enumerate(maps):
            out[f"{name}.train.map{i}.summary"] = summary(mp)
            out[f"{name}.train.map{i}.samples"] = samples(mp)
            out[f"{name}.train.map{i}.shape"] = np.array(mp.shape)
        filler.fill_module(model, name + ".")  # undo the running-stat update of the probe above
        logits = model(x)
        loss = F.cross_entropy(logits, y, label_smoothing=LABEL_SMOOTHING)
        loss.backward()
        out[f"{name}.train.logits"] = np_(logits)
        out[f"{name}.train.loss"] = np.array(loss.item())
        keys, norms = [], []
        for k, p in model.named_parameters():
            keys.append(k)
            norms.append(p.grad.double().norm().item())
        out[f"{name}.train.grad_keys"] = np.array(keys)
        out[f"{name}.train.grad_norms"] = np.array(norms)
        first_bn = next(k for k in model.state_dict() if k.endswith("running_mean"))
        last_bn = [k for k in model.state_dict() if k.endswith("running_var")][-1]
        out[f"{name}.train.first_running_mean"] = np_(model.state_dict()[first_bn])
        out[f"{name}.train.last_running_var"] = np_(model.state_dict()[last_bn])
        # a few full gradients
        sd_grads = dict(model.named_parameters())
        for k in (keys[0], keys[len(keys) // 2], "3.weight", "3.bias"):
            out[f"{name}.train.grad.{k}"] = np_(sd_grads[k].grad)
        # eval-mode forward WITH gradients (BatchNorm uses its running statistics as constants):
        # a well-conditioned check of the whole backward wiring, free of the batch-statistics
        # amplification that 16 samples per channel cause in train mode
        filler.fill_module(model, name + ".")
        model.eval()
        model.zero_grad()
        logits = model(x)
        loss = F.cross_entropy(logits, y, label_smoothing=LABEL_SMOOTHING)
        loss.backward()
        out[f"{name}.evalgrad.loss"] = np.array(loss.item())
        out[f"{name}.evalgrad.grad_norms"] = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
        sd_grads = dict(model.named_parameters())
        for k in (keys[0], keys[1], keys[len(keys) // 2], "3.weight"):
            out[f"{name}.evalgrad.grad.{k}"] = np_(sd_grads[k].grad)
        # eval forward
        filler.fill_module(model, name + ".")
        model.eval()
        with torch.no_grad():
            maps = model[0].get_feature_maps(x)
            for i, mp in enumerate(maps):
                out[f"{name}.eval.map{i}.summary"] = summary(mp)
                out[f"{name}.eval.map{i}.samples"] = samples(mp)
            out[f"{name}.eval.logits"] = np_(model(x))
        print("model", name, "loss", loss.item())
    # BASELINE config 1: Darknet-19 forward, 1x3x224x224
    model = make_classifier("darknet19")
    filler.fill_module(model, "darknet19.")
    model.eval()
    x = filler.images(1, 224, seed=224)
    with torch.no_grad():
        f = model[0](x)
        out["darknet19.cfg1.last.summary"] = summary(f)
        out["darknet19.cfg1.last.samples"] = samples(f, 256)
        out["darknet19.cfg1.logits"] = np_(model(x))
    # BASELINE config 5: Darknet-YOLOv5x get_feature_maps() @640px.  Eval mode (running statistics), so an
    # image's maps do not depend on its batch mates: the GPU test places these two images at positions 0
    # and 63 of its 64-image batch.
    model = FACTORIES["darknet_yolov5x"]()
    filler.fill_module(model, "darknet_yolov5x.cfg5.")
    model.eval()
    x = filler.images(2, 640, seed=640)
    with torch.no_grad():
        maps = model.get_feature_maps(x)
    for i, mp in enumerate(maps):
        out[f"darknet_yolov5x.cfg5.map{i}.shape"] = np.array(mp.shape)
        for b in range(2):
            out[f"darknet_yolov5x.cfg5.map{i}.img{b}.summary"] = summary(mp[b])
            out[f"darknet_yolov5x.cfg5.map{i}.img{b}.samples"] = samples(mp[b], 512)
    np.savez_compressed(GOLDEN / "models.npz", **out)
    print("models:", len(out), "arrays")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    GOLDEN.mkdir(parents=True, exist_ok=True)
    gen_manifest()
    gen_units()
    gen_blocks()
    gen_models()
    for f in sorted(GOLDEN.iterdir()):
        print(f.name, f.stat().st_size)
