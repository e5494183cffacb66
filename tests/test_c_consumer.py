"""The drop-in boundary from C: tests/c_consumer/consumer.c uses include/vt_amd.h, the HIP runtime and nothing else (no Python,
no PyTorch) to run one training-mode ConvNormAct unit (reference components.py:26-44) through libvt_amd.so, and checks it
against the plain-C oracle (oracle/ref_ops.c).  Without a GPU: it must compile as C99 and link (every entry point it calls is
exported with the declared signature).  On the GPU box: it must print C_CONSUMER_OK."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "c_consumer" / "consumer.c"
LIB = ROOT / "vision-toolbox_amd" / "csrc" / "libvt_amd.so"
REF = ROOT / "oracle" / "libvt_ref.so"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))


def _build(out: Path) -> Path:
    if not LIB.exists() or not REF.exists():
        pytest.skip("libvt_amd.so / libvt_ref.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    if shutil.which("gcc") is None or not (ROCM / "include" / "hip" / "hip_runtime_api.h").exists():
        pytest.skip("gcc or the HIP headers are missing")
    exe = out / "consumer"
    cmd = ["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__",
           f"-I{ROCM / 'include'}", f"-I{ROOT / 'include'}", str(SRC), str(LIB), str(REF), f"-L{ROCM / 'lib'}", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{ROCM / 'lib'}", f"-Wl,-rpath,{LIB.parent}", f"-Wl,-rpath,{REF.parent}", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_consumer_compiles_and_links_as_c99(tmp_path):
    exe = _build(tmp_path)
    assert exe.exists() and os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c_consumer_runs_a_conv_norm_act_unit_against_the_c_oracle(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_CONSUMER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
