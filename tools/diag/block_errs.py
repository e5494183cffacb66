"""Dev (GPU box): per-tensor errors of the batch-256 block tests (tests/test_fullsize_gpu.py) -- python tools/diag/block_errs.py [csp|osa]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]
import conftest  # noqa: F401  (test defaults: pointwise threshold)
import torch

import test_fullsize_gpu as T

which = sys.argv[1] if len(sys.argv) > 1 else "csp"
orig = T._block_case


def show(m, x, ref):
    errs = orig(m, x, ref)
    for k, (rel, slope, n, floor) in errs.items():
        print(f"  {k:28s} rel {rel:.3e}  slope-1 {slope - 1:+.2e}  n {n:8d}  emulating-ref vs float64 {floor:.3e}")
    return errs


T._block_case = show
if which == "csp":
    T.test_csp_stage_bf16_train_mode_gradients_at_batch_256_are_tight()
else:
    T.test_osa_block_bf16_train_mode_gradients_at_batch_256_are_tight()
