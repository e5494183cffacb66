"""Checkpoint interop with the reference.

The on-disk contract is the reference's: a flat state dict with the reference's keys, OIHW float32
conv weights (backbones/base.py:23-25 loads it with `load_state_dict`).  Internally this build keeps
the filters channels_last inside one flat buffer (engine.ParamStore); everything here converts at the
file boundary only.

* `reference_state_dict(module)`   -> CPU, contiguous, float32 / int64 tensors under the reference's keys
* `save_backbone(module, path)`    -> the file `_load_state_dict_from_url` / the reference would load
* `extract_backbone_weights(...)`  -> the reference's extras.py:112-128: strip `model.0.` from a Lightning
                                      classifier checkpoint, name the file `<save_name>-<sha256[:8]>.pth`
* `toolbox_to_yolov5_key` / `yolov5_to_toolbox_key` / `convert_yolov5_weights`
                                   -> the key grammar of scripts/convert_yolov5_weights.py:6-52 and its inverse
"""
from __future__ import annotations

import hashlib
import io
import os
import re
from typing import Dict

import torch

__all__ = ["reference_state_dict", "save_backbone", "extract_backbone_weights", "toolbox_to_yolov5_key",
           "yolov5_to_toolbox_key", "convert_yolov5_weights"]


def reference_state_dict(module: torch.nn.Module) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in module.state_dict().items():
        t = v.detach().to("cpu")
        out[k] = t.contiguous().clone() if t.dim() else t.clone()
    return out


def save_backbone(module: torch.nn.Module, path) -> None:
    torch.save(reference_state_dict(module), path)


def extract_backbone_weights(lightning_ckpt_path, save_name: str, save_dir=None) -> str:
    """the backbone of a classifier checkpoint (`model.0.*`, classifier.py:59) as a release file"""
    save_dir = os.getcwd() if save_dir is None else save_dir
    ckpt = torch.load(lightning_ckpt_path, map_location="cpu")
    prefix = "model.0."
    weights = {k[len(prefix):]: v for k, v in ckpt["state_dict"].items() if k.startswith(prefix)}
    buf = io.BytesIO()
    torch.save(weights, buf)
    data = buf.getvalue()
    path = os.path.join(save_dir, f"{save_name}-{hashlib.sha256(data).hexdigest()[:8]}.pth")
    with open(path, "wb") as f:
        f.write(data)
    return path


# YOLOv5's backbone numbers its layers: 0 = stem, then (conv, C3) pairs: stage s -> layers 2s+1, 2s+2.
# Inside C3: cv1 feeds the bottlenecks (`m`), cv2 is the bypass, cv3 the output conv -- i.e. this build's
# conv2 / conv1 / out_conv (CSPDarknetStage: conv1 is the bypass).
_STAGE_SUB = {"conv1": "cv2", "conv2": "cv1", "out_conv": "cv3"}
_C3_SUB = {v: k for k, v in _STAGE_SUB.items()}


def toolbox_to_yolov5_key(key: str) -> str:
    parts = key.split(".")
    if parts[0] == "stem":
        return ".".join(["model", "0"] + parts[1:])
    if parts[0] != "stages" or len(parts) < 4:
        raise ValueError(f"Unexpected weight name: {key}")
    stage, sub, rest = int(parts[1]), parts[2], parts[3:]
    if sub == "conv":
        return ".".join(["model", str(2 * stage + 1)] + rest)
    if sub in _STAGE_SUB:
        return ".".join(["model", str(2 * stage + 2), _STAGE_SUB[sub]] + rest)
    if sub == "blocks":
        rest = [rest[0], rest[1].replace("conv", "cv")] + rest[2:]
        return ".".join(["model", str(2 * stage + 2), "m"] + rest)
    raise ValueError(f"Unexpected weight name: {key}")


def yolov5_to_toolbox_key(key: str) -> str:
    parts = key.split(".")
    if parts[0] != "model" or len(parts) < 3:
        raise ValueError(f"Unexpected weight name: {key}")
    idx, rest = int(parts[1]), parts[2:]
    if idx == 0:
        return ".".join(["stem"] + rest)
    stage = (idx - 1) // 2
    if idx % 2 == 1:
        return ".".join(["stages", str(stage), "conv"] + rest)
    if rest[0] in _C3_SUB:
        return ".".join(["stages", str(stage), _C3_SUB[rest[0]]] + rest[1:])
    if rest[0] == "m" and re.fullmatch(r"cv[12]", rest[2]):
        return ".".join(["stages", str(stage), "blocks", rest[1], rest[2].replace("cv", "conv")] + rest[3:])
    raise ValueError(f"Unexpected weight name: {key}")


def convert_yolov5_weights(src_path, dst_path, to: str = "yolov5", verbose: bool = True) -> None:
    """re-key a checkpoint file (scripts/convert_yolov5_weights.py: toolbox keys -> YOLOv5 layer numbering;
    to='toolbox' is the inverse)"""
    fn = {"yolov5": toolbox_to_yolov5_key, "toolbox": yolov5_to_toolbox_key}[to]
    weights = torch.load(src_path, map_location="cpu")
    out = {}
    for k, v in weights.items():
        nk = fn(k)
        out[nk] = v
        if verbose:
            print(f"{k} -> {nk}. Shape: {tuple(v.shape)}")
    torch.save(out, dst_path)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("src_path")
    ap.add_argument("dst_path")
    ap.add_argument("--to", choices=["yolov5", "toolbox"], default="yolov5")
    a = ap.parse_args()
    convert_yolov5_weights(a.src_path, a.dst_path, a.to)
