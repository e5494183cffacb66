"""Isolated timing (GPU box): a raw 1x1 convolution as the pointwise apply pass with unit coefficients (scale 1, shift 0, no
ReLU: y = bf16(W x), the filter in registers, no barrier in the loop) against vt_conv_igemm on the same operands.

    python tools/bench_pw_as_conv.py"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tests")]

import torch

from vision_toolbox import _native as N
from gpu_util import conv_desc


def timeit(fn, iters=60, warmup=10):
    s = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(warmup):
        fn(s)
    e0, e1 = N.Event(), N.Event()
    e0.record(s)
    for _ in range(iters):
        fn(s)
    e1.record(s)
    return e0.elapsed_ms(e1) / iters * 1e3


def main():
    lib = N.lib()
    vp = lambda t: C.c_void_p(t.data_ptr())
    for B, HW, K, Nn in [(256, 28, 128, 128), (256, 56, 128, 128), (256, 56, 128, 64), (256, 56, 64, 64), (256, 14, 128, 128)]:
        M = B * HW * HW
        x4 = torch.randn(B, HW, HW, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Nn, K, device="cuda") * (2.0 / K) ** 0.5).to(torch.bfloat16)
        y0 = torch.empty(B, HW, HW, Nn, device="cuda", dtype=torch.bfloat16)
        y1 = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
        coef = torch.zeros(4, Nn, device="cuda")
        coef[0] = 1.0
        d = conv_desc(N.VT_BF16, x4, K, Nn, 1, 1, 0, Nn, 0)
        pd = N.PwDesc()
        pd.dtype, pd.K, pd.ngroups, pd.relu, pd.M = N.VT_BF16, K, 1, 0, M
        pd.x, pd.ldx = x4.data_ptr(), K
        pd.C[0], pd.w[0], pd.ldw[0] = Nn, w.data_ptr(), K
        ys = (C.c_void_p * 1)(y1.data_ptr())
        ld = (C.c_int32 * 1)(Nn)
        none = (C.c_void_p * 1)(None)
        zero = (C.c_int32 * 1)(0)

        def conv(st):
            N.check(lib.vt_conv_igemm(C.byref(d), vp(x4), vp(w), vp(y0), None, None, None, None, st))

        def pw(st):
            N.check(lib.vt_pw_fwd_apply(C.byref(pd), coef.data_ptr(), ys, ld, none, zero, st))

        tc = min(timeit(conv), timeit(conv))
        kc = N.last_kernel_name()
        tp = min(timeit(pw), timeit(pw))
        torch.cuda.synchronize()
        eq = torch.equal(y0.view(M, Nn), y1)
        nb = 2.0 * M * (K + Nn)
        print(f"{K:4d}->{Nn:4d} @{HW:3d}x{HW:<3d}: conv {tc:7.1f} us ({nb / tc / 1e6:5.2f} TB/s) [{kc}] | pointwise apply, unit coefficients {tp:7.1f} us "
              f"({nb / tp / 1e6:5.2f} TB/s) | outputs {'EQUAL' if eq else 'differ: %.3g of the elements' % (y0.view(M, Nn) != y1).float().mean().item()}", flush=True)


main()
