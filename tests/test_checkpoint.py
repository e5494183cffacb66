"""Checkpoint interop: the YOLOv5 key grammar of scripts/convert_yolov5_weights.py:6-52,
extras.extract_backbone_weights (extras.py:112-128), and save -> reference format -> load round trips."""
import hashlib
import importlib.util
import json
import sys
from pathlib import Path

import pytest
import torch

from oracle import filler
from vision_toolbox import backbones, checkpoint

REF = Path("/root/reference")


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


def _reference_converter():
    spec = importlib.util.spec_from_file_location("ref_convert", REF / "scripts" / "convert_yolov5_weights.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("variant", ["n", "x"])
def test_yolov5_key_grammar(variant, manifest, tmp_path):
    keys = [k for k, _ in manifest[f"darknet_yolov5{variant}"]["keys"]]
    mapped = [checkpoint.toolbox_to_yolov5_key(k) for k in keys]
    assert len(set(mapped)) == len(keys)
    assert [checkpoint.yolov5_to_toolbox_key(k) for k in mapped] == keys
    # the documented rules (convert_yolov5_weights.py:10-16)
    rules = {"stem.conv.weight": "model.0.conv.weight", "stages.0.conv.norm.bias": "model.1.norm.bias",
             "stages.0.conv1.conv.weight": "model.2.cv2.conv.weight", "stages.0.conv2.conv.weight": "model.2.cv1.conv.weight",
             "stages.1.blocks.1.conv2.norm.weight": "model.4.m.1.cv2.norm.weight",
             "stages.3.out_conv.conv.weight": "model.8.cv3.conv.weight"}
    for k, v in rules.items():
        assert checkpoint.toolbox_to_yolov5_key(k) == v and checkpoint.yolov5_to_toolbox_key(v) == k
    with pytest.raises(ValueError, match="Unexpected weight name"):
        checkpoint.toolbox_to_yolov5_key("head.weight")
    if REF.exists():  # build container only: the unmodified script must produce the same file
        m = getattr(backbones, f"darknet_yolov5{variant}")()
        src, a, b = tmp_path / "src.pth", tmp_path / "a.pth", tmp_path / "b.pth"
        checkpoint.save_backbone(m, src)
        _reference_converter().convert_weights(str(src), str(a))
        checkpoint.convert_yolov5_weights(src, b, verbose=False)
        sa, sb = torch.load(a), torch.load(b)
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
        checkpoint.convert_yolov5_weights(b, tmp_path / "c.pth", to="toolbox", verbose=False)
        sc, ss = torch.load(tmp_path / "c.pth"), torch.load(src)
        assert list(sc) == list(ss) and all(torch.equal(sc[k], ss[k]) for k in ss)


def test_extract_backbone_weights(tmp_path):
    bb = backbones.vovnet19_slim_ese()
    model = torch.nn.Sequential(bb, torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(512, 10))
    filler.fill_module(model, "ckpt.")
    lightning = {"state_dict": {"model." + k: v for k, v in model.state_dict().items()}, "epoch": 3}
    torch.save(lightning, tmp_path / "last.ckpt")
    path = checkpoint.extract_backbone_weights(tmp_path / "last.ckpt", "vovnet19_slim_ese", tmp_path)
    data = Path(path).read_bytes()
    assert Path(path).name == f"vovnet19_slim_ese-{hashlib.sha256(data).hexdigest()[:8]}.pth"
    sd = torch.load(path)
    assert list(sd) == list(bb.state_dict()) and not any(k.startswith(("1.", "3.")) for k in sd)
    fresh = backbones.vovnet19_slim_ese()
    fresh.load_state_dict(sd)
    x = filler.images(1, 64)
    bb.eval(), fresh.eval()
    with torch.no_grad():
        torch.testing.assert_close(fresh(x), bb(x))


@pytest.mark.skipif(not REF.exists(), reason="the reference checkout exists only in the build container")
@pytest.mark.parametrize("name", ["cspdarknet53", "darknet_yolov5n", "vovnet19_slim_ese"])
def test_saved_checkpoint_loads_into_the_unmodified_reference(name, tmp_path):
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import gen_golden as G  # imports the unmodified reference behind the torchvision shim

    ours = getattr(backbones, name)()
    filler.fill_module(ours, "rt.")
    ours._vt_runner().store.ensure(torch.device("cpu"))  # flat channels_last storage, as after a GPU run
    path = tmp_path / f"{name}.pth"
    checkpoint.save_backbone(ours, path)
    sd = torch.load(path)
    assert all(v.is_contiguous() for v in sd.values())
    ref = G.FACTORIES[name]()
    ref.load_state_dict(sd)  # strict: same keys, same OIHW shapes
    x = filler.images(2, 64)
    ours.eval(), ref.eval()
    with torch.no_grad():
        a, b = ours.get_feature_maps(x), ref.get_feature_maps(x)
    for u, v in zip(a, b):
        assert ((u - v).norm() / v.norm()).item() < 1e-5  # (channels_last filter strides pick another ATen kernel)
    # and back: a file written by the reference loads here
    torch.save(ref.state_dict(), tmp_path / "ref.pth")
    again = getattr(backbones, name)()
    again.load_state_dict(torch.load(tmp_path / "ref.pth"))
    with torch.no_grad():
        assert ((again.eval()(x) - b[-1]).norm() / b[-1].norm()).item() < 1e-5


@pytest.mark.gpu
def test_gpu_module_round_trips_through_the_reference_format(tmp_path):
    """train one step on the GPU (weights, running statistics and counters change inside the flat store),
    save in the reference's format, load into a fresh module and into the CPU oracle"""
    from oracle import torch_ref as R

    m = backbones.darknet_yolov5n()
    filler.fill_module(m, "rt.")
    m = m.cuda().train()
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    x = filler.images(4, 64).cuda()
    m(x).square().mean().backward()
    opt.step()
    path = tmp_path / "y5n.pth"
    checkpoint.save_backbone(m, path)
    sd = torch.load(path)
    assert all(v.device.type == "cpu" and v.is_contiguous() for v in sd.values())
    assert int(sd["stem.norm.num_batches_tracked"]) == 1
    m.eval()
    with torch.no_grad():
        got = m.get_feature_maps(x)
        ref = R.feature_maps("darknet_yolov5n", {k: v.clone() for k, v in sd.items()}, x.cpu(), False)
    for a, b in zip(got, ref):
        assert ((a.float().cpu() - b).norm() / b.norm()).item() < 1e-3
    fresh = backbones.darknet_yolov5n()
    fresh.load_state_dict(sd)
    with torch.no_grad():
        torch.testing.assert_close(fresh.cuda().eval()(x), got[-1], rtol=1e-4, atol=1e-4)
