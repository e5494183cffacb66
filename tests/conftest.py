"""pytest configuration: markers and import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI export check (CPU only).
`-m gpu`: parity tests proper; they call the HIP library through the C-ABI.
"""
import os
import sys
from pathlib import Path

import pytest

# The pointwise (1x1 recompute) kernels are taken by default only where a unit's input is >= 80 MB (engine.py,
# VT_PW_MIN_MB): the toy shapes of the model-level parity tests would never reach them.  The tests lower the threshold,
# so every model-level test exercises the pointwise path wherever a kernel exists; the unfused path keeps its own
# kernel-level tests and runs in every unit the pointwise kernels do not cover (3x3, f32, odd channel counts).
os.environ.setdefault("VT_PW_MIN_MB", "0")

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "vision-toolbox_amd"
for p in (str(PKG), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
