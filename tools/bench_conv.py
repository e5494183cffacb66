"""Micro-benchmark of the conv kernels on CSPDarknet-53 layer shapes (GPU box).

    VT_IGEMM_VARIANT=k python tools/bench_conv.py [fwd|wgrad|all]

Prints ms and TFLOP/s per layer for the implicit-GEMM forward kernel (with the training
epilogue) and the filter-gradient kernel, timed with HIP events on the launch stream."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N

LAYERS = [  # (Cin, Cout, k, s, Hin) at batch 256
    (128, 128, 3, 1, 28),
    (256, 256, 3, 1, 14),
    (512, 512, 3, 1, 7),
    (64, 64, 3, 1, 56),
    (128, 256, 3, 2, 56),
    (256, 256, 1, 1, 28),
    (512, 512, 1, 1, 14),
    (64, 128, 3, 2, 112),
    (32, 64, 3, 2, 224),
]


def desc_for(B, Cin, Cout, k, s, H, flags):
    pad = -((s - k) // 2)
    Ho = (H + 2 * pad - k) // s + 1
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, H, Cin, Cin
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = Ho, Ho, s, s, -pad, -pad
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, Cout, Ho, Ho, 1, 1
    d.ldw, d.flags, d.ntaps = k * k * Cin, flags, k * k
    for i in range(k * k):
        d.dh[i], d.dw[i] = i // k, i % k
    return d, Ho


def timeit(fn, iters=20, warmup=5):
    s = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(warmup):
        fn(s)
    e0, e1 = N.Event(), N.Event()
    e0.record(s)
    for _ in range(iters):
        fn(s)
    e1.record(s)
    return e0.elapsed_ms(e1) / iters


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B = int(os.environ.get("VT_BENCH_BATCH", "256"))
    lib = N.lib()
    print(f"variant={os.environ.get('VT_IGEMM_VARIANT', '0')} wvariant={os.environ.get('VT_WGRAD_VARIANT', '0')} B={B}")
    layers = LAYERS
    if len(sys.argv) > 2:  # custom layers: Cin,Cout,k,s,H ...
        layers = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
    for Cin, Cout, k, s, H in layers:
        aff = bool(os.environ.get("VT_BENCH_AFFINE"))  # the inference epilogue (folded BatchNorm + ReLU [+ residual])
        res = bool(os.environ.get("VT_BENCH_RESIDUAL"))
        fl = (N.VT_CONV_AFFINE | N.VT_CONV_RELU | (N.VT_CONV_RESIDUAL if res else 0)) if aff else (
            N.VT_CONV_RESIDUAL if res else (0 if os.environ.get("VT_BENCH_NOSTATS") else N.VT_CONV_STATS))  # res alone: a data gradient's accumulate
        d, Ho = desc_for(B, Cin, Cout, k, s, H, fl)
        d.ldr = Cout
        sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
        x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Cout, k * k, Cin, device="cuda") * (2.0 / (k * k * Cin)) ** 0.5).to(torch.bfloat16)
        if os.environ.get("VT_BENCH_ZERO"):  # DVFS probe: the same launches on all-zero operands (guide: give-back item 1)
            x.zero_()
            w.zero_()
        y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=torch.bfloat16)
        stats = N.stats_buffer(Cout)
        dz = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16)
        dw = torch.zeros(Cout, k * k, Cin, device="cuda")
        flops = 2.0 * B * Ho * Ho * Cout * k * k * Cin
        line = f"{Cin:4d}->{Cout:4d} k{k} s{s} @{H:3d}  {flops / 1e9:7.1f} GF"
        if what in ("fwd", "all"):
            resid = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16) if res else None
            ms = timeit(lambda st: N.check(lib.vt_conv_igemm(
                ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), sc.data_ptr() if aff else None,
                sf.data_ptr() if aff else None, resid.data_ptr() if res else None, None if (aff or res) else stats.data_ptr(), st)))
            line += f" | fwd {ms:7.4f} ms {flops / ms / 1e9:7.1f} TF/s [{N.last_kernel_name()}]"
        if what == "bnred":  # vt_conv_dgrad_bnred against the two launches it replaces (flags 0 launch + the reduction pass)
            d0, _ = desc_for(B, Cin, Cout, k, s, H, 0)
            z = (torch.randn(B, Ho, Ho, Cout, device="cuda") + 0.2).to(torch.bfloat16)
            mean, istd = torch.randn(Cout, device="cuda") * 0.1 + 0.2, torch.rand(Cout, device="cuda") + 0.5
            sums = N.stats_buffer(Cout)
            vp_ = lambda t: ctypes.c_void_p(t.data_ptr())
            ms_f = timeit(lambda st: N.check(lib.vt_conv_dgrad_bnred(ctypes.byref(d0), vp_(x), vp_(w), vp_(y), vp_(z), Cout, vp_(sc), vp_(sf),
                                                                     vp_(mean), vp_(istd), 1, vp_(sums), st)))
            name = N.last_kernel_name()
            ms_c = timeit(lambda st: N.check(lib.vt_conv_igemm(ctypes.byref(d0), vp_(x), vp_(w), vp_(y), None, None, None, None, st)))
            ms_r = timeit(lambda st: N.check(lib.vt_bn_act_bwd_reduce(vp_(y), Cout, vp_(z), Cout, vp_(sc), vp_(sf), vp_(mean), vp_(istd),
                                                                      B * Ho * Ho, Cout, 1, N.VT_BF16, vp_(sums), st)))
            line += f" | fused {ms_f * 1e3:7.1f} us [{name}] | plain {ms_c * 1e3:7.1f} us + reduce {ms_r * 1e3:6.1f} us"
        if what in ("wgrad", "all"):
            d0, _ = desc_for(B, Cin, Cout, k, s, H, 0)
            ms = timeit(lambda st: N.check(lib.vt_conv_wgrad(ctypes.byref(d0), x.data_ptr(), dz.data_ptr(),
                                                             dw.data_ptr(), k * k * Cin, st)))
            line += f" | wgrad {ms:7.4f} ms {flops / ms / 1e9:7.1f} TF/s [{N.last_kernel_name()}]"
            G = int(os.environ.get("VT_BENCH_GROUP", "0"))
            if G > 1:  # G same-shape layers through vt_conv_wgrad_group (operands of their own: no cache help)
                xs = [x] + [torch.randn_like(x) for _ in range(G - 1)]
                dzs = [dz] + [torch.randn_like(dz) for _ in range(G - 1)]
                dws = [dw] + [torch.zeros_like(dw) for _ in range(G - 1)]
                arr = lambda ts: (ctypes.c_void_p * G)(*[t.data_ptr() for t in ts])
                ax, az, aw = arr(xs), arr(dzs), arr(dws)
                ms = timeit(lambda st: N.check(lib.vt_conv_wgrad_group(ctypes.byref(d0), G, ax, az, aw, k * k * Cin, st)))
                line += f" | group of {G}: {ms / G:7.4f} ms per layer {flops * G / ms / 1e9:7.1f} TF/s [{N.last_kernel_name()}]"
        print(line, flush=True)


if __name__ == "__main__":
    main()
