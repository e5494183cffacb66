"""The plain-C oracle (oracle/ref_ops.c, double accumulation, no torch) against the torch CPU
arithmetic the reference runs on -- two independent restatements must agree."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import filler

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def ref():
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return C.CDLL(str(ROOT / "oracle" / "libvt_ref.so"))


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("case", [(2, 5, 7, 6, 4, 3, 1), (2, 4, 9, 9, 6, 3, 2), (1, 3, 12, 12, 8, 6, 2), (2, 8, 5, 5, 8, 1, 1)],
                         ids=str)
def test_conv2d_forward_backward(ref, case):
    B, Cin, H, W, Cout, k, s = case
    pad = -((s - k) // 2)
    x = filler.tensor(f"cx{case}", (B, Cin, H, W)).requires_grad_(True)
    w = filler.tensor(f"cw{case}", (Cout, Cin, k, k)).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), None, s, pad)
    dy = filler.tensor(f"cdy{case}", y.shape)
    y.backward(dy.double())
    xn, wn, dyn = x.detach().numpy(), w.detach().numpy(), dy.numpy()
    yn = np.zeros(tuple(y.shape), np.float32)
    ref.vt_ref_conv2d_fwd(fp(xn), fp(wn), fp(yn), B, Cin, H, W, Cout, k, s, pad)
    np.testing.assert_allclose(yn, y.detach().numpy(), rtol=1e-5, atol=1e-6)
    dx, dw = np.zeros_like(xn), np.zeros_like(wn)
    ref.vt_ref_conv2d_bwd(fp(xn), fp(wn), fp(dyn), fp(dx), fp(dw), B, Cin, H, W, Cout, k, s, pad)
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw, w.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_relu(ref, training):
    B, Cc, H, W = 3, 6, 5, 4
    z = (filler.tensor("bz", (B, Cc, H, W)) * 2 + 0.5).requires_grad_(True)
    gamma = (filler.tensor("bg", (Cc,)) * 0.3 + 1).requires_grad_(True)
    beta = (filler.tensor("bb", (Cc,)) * 0.3).requires_grad_(True)
    rm, rv = filler.tensor("brm", (Cc,)) * 0.1, filler.tensor("brv", (Cc,)).abs() + 0.5
    rm_t, rv_t = rm.clone(), rv.clone()
    y = torch.relu(F.batch_norm(z, rm_t, rv_t, gamma, beta, training, 0.1, 1e-5))
    rmn, rvn = rm.numpy().copy(), rv.numpy().copy()
    yn = np.zeros((B, Cc, H, W), np.float32)
    ref.vt_ref_bn_relu_fwd(fp(z.detach().numpy()), fp(gamma.detach().numpy()), fp(beta.detach().numpy()), fp(rmn),
                           fp(rvn), fp(yn), B, Cc, H * W, C.c_float(1e-5), C.c_float(0.1), int(training), 1)
    np.testing.assert_allclose(yn, y.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rmn, rm_t.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvn, rv_t.numpy(), rtol=1e-5, atol=1e-6)
    if training:
        dy = filler.tensor("bdy", y.shape)
        y.backward(dy)
        dz, dg, db = np.zeros((B, Cc, H, W), np.float32), np.zeros(Cc, np.float32), np.zeros(Cc, np.float32)
        ref.vt_ref_bn_relu_bwd(fp(z.detach().numpy()), fp(gamma.detach().numpy()), fp(beta.detach().numpy()),
                               fp(dy.numpy()), fp(dz), fp(dg), fp(db), B, Cc, H * W, C.c_float(1e-5), 1)
        np.testing.assert_allclose(dz, z.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dg, gamma.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(db, beta.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_maxpool_first_maximum_wins(ref):
    B, Cc, H, W = 2, 3, 9, 8
    x = torch.relu(filler.tensor("mx", (B, Cc, H, W))).requires_grad_(True)  # many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = filler.tensor("mdy", y.shape)
    y.backward(dy)
    yn = np.zeros(tuple(y.shape), np.float32)
    am = np.zeros(tuple(y.shape), np.int32)
    ref.vt_ref_maxpool3x3s2_fwd(fp(x.detach().numpy()), fp(yn), fp(am), B, Cc, H, W)
    np.testing.assert_array_equal(yn, y.detach().numpy())
    dx = np.zeros((B, Cc, H, W), np.float32)
    ref.vt_ref_maxpool3x3s2_bwd(fp(dy.numpy()), fp(am), fp(dx), B, Cc, H, W)
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-6, atol=1e-6)
