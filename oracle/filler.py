"""TEST INFRASTRUCTURE -- deterministic, version-independent tensor filler.

Every tensor is generated from a numpy legacy RandomState seeded by the CRC32 of its
state_dict key, so the reference (when the golden vectors are generated), the oracle and
the HIP build are given bit-identical weights without relying on torch's RNG stream.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rs(key: str) -> np.random.RandomState:
    return np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)


def fill_tensor(key: str, t: torch.Tensor) -> torch.Tensor:
    """value for the state_dict entry `key` with the shape / dtype of `t`."""
    rs = _rs(key)
    shape = tuple(t.shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=t.dtype)
    n = rs.standard_normal(shape).astype(np.float64)
    if leaf == "running_mean":
        v = 0.1 * n
    elif leaf == "running_var":
        v = 1.0 + 0.25 * np.abs(n)
    elif key.endswith("norm.weight"):
        v = 1.0 + 0.1 * n
    elif key.endswith("norm.bias") or leaf == "bias":
        v = 0.1 * n
    elif leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        v = n * np.sqrt(2.0 / fan_in)
    elif leaf == "weight" and len(shape) == 2:
        v = n * np.sqrt(1.0 / shape[1])
    else:
        v = 0.1 * n
    return torch.from_numpy(v.astype(np.float32)).to(t.dtype)


def fill_state_dict(sd: dict, prefix: str = "") -> dict:
    return {k: fill_tensor(prefix + k, v) for k, v in sd.items()}


def fill_module(module: torch.nn.Module, prefix: str = "") -> None:
    """overwrite every parameter and buffer of `module` in place."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(fill_tensor(prefix + k, v))


def images(batch: int, size: int, seed: int = 1234, channels: int = 3) -> torch.Tensor:
    """uniform [0,1) images like tests/test_backbones.py:21 of the reference."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.rand(batch, channels, size, size).astype(np.float32))


def labels(batch: int, num_classes: int, seed: int = 4321) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randint(0, num_classes, size=(batch,)).astype(np.int64))


def tensor(key: str, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((_rs(key).standard_normal(tuple(shape)) * scale).astype(np.float32))
