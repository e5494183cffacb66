"""Micro-benchmark of the HBM-bound kernels on CSPDarknet-53 activation shapes (GPU box).

    python tools/bench_eltwise.py

Prints ms and achieved GB/s (algorithmic bytes: every operand read/written once) for the
BatchNorm apply / backward-reduce / backward-apply kernels at batch 256."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N

SHAPES = [(32, 224), (64, 112), (64, 56), (128, 56), (128, 28), (256, 28), (256, 14), (512, 14), (512, 7), (1024, 7)]


def timeit(fn, iters=20, warmup=3):
    """GPU time per launch: the launches are captured into one hipGraph (the Python/ctypes
    launch cost, ~25 us, would otherwise floor the small tensors)."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = int(st.cuda_stream)
        for _ in range(warmup):
            fn(s)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn(int(torch.cuda.current_stream().cuda_stream))
        g.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        g.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(os.environ.get("VT_BENCH_BATCH", "256"))
    lib = N.lib()
    dev = torch.device("cuda")
    tot = {"apply": 0.0, "reduce": 0.0, "bwd_apply": 0.0}
    for C, H in SHAPES:
        M = B * H * H
        z = torch.randn(M, C, device=dev).to(torch.bfloat16)
        dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
        y = torch.empty_like(z)
        scale = torch.rand(C, device=dev) + 0.5
        shift = torch.randn(C, device=dev) * 0.1
        mean = torch.randn(C, device=dev) * 0.1
        invstd = torch.rand(C, device=dev) + 0.5
        sums = N.stats_buffer(C)
        coef = torch.rand(3, C, device=dev)
        vp = ctypes.c_void_p

        def f_apply(s):
            N.check(lib.vt_bn_act_apply(vp(z.data_ptr()), C, vp(scale.data_ptr()), vp(shift.data_ptr()), None, 0,
                                        vp(y.data_ptr()), C, M, C, 1, N.VT_BF16, vp(s)))

        def f_reduce(s):
            N.check(lib.vt_bn_act_bwd_reduce(vp(dy.data_ptr()), C, vp(z.data_ptr()), C, vp(scale.data_ptr()),
                                             vp(shift.data_ptr()), vp(mean.data_ptr()), vp(invstd.data_ptr()), M, C, 1,
                                             N.VT_BF16, vp(sums.data_ptr()), vp(s)))

        def f_bapply(s):
            N.check(lib.vt_bn_act_bwd_apply(vp(dy.data_ptr()), C, vp(z.data_ptr()), C, vp(scale.data_ptr()),
                                            vp(shift.data_ptr()), vp(coef.data_ptr()), vp(y.data_ptr()), C, M, C, 1,
                                            N.VT_BF16, vp(s)))

        nb = M * C * 2
        row = f"C={C:5d} @{H:3d}  {nb/1e6:8.1f} MB/tensor |"
        timeit(f_bapply)  # settle clocks / caches before the first measured kernel
        for name, fn, ntens in (("apply", f_apply, 2), ("reduce", f_reduce, 2), ("bwd_apply", f_bapply, 3)):
            ms = timeit(fn)
            tot[name] += ms
            row += f" {name} {ms:7.4f} ms {ntens*nb/ms/1e6:7.0f} GB/s |"
        print(row)
    print("sum ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
