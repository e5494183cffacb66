"""Isolated timing (GPU box): vt_bn_act_bwd_reduce_finalize (the finalize step as the reduction's tail) against the two
launches it replaces, each followed by the apply pass that consumes the coefficients; CSPDarknet-53 shapes at batch 256.

    python tools/bench_fin_tail.py"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N


def timeit(fn, iters=40, warmup=8):
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
    for M, Cc in [(256 * 112 * 112, 64), (256 * 56 * 56, 64), (256 * 28 * 28, 128), (256 * 14 * 14, 256), (256 * 7 * 7, 512),
                  (256 * 7 * 7, 1024)]:
        dy = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
        z = (torch.randn(M, Cc, device="cuda") + 0.3).to(torch.bfloat16)
        dz = torch.empty_like(dy)
        scale, shift = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
        mean, invstd = torch.randn(Cc, device="cuda") * 0.1 + 0.3, torch.rand(Cc, device="cuda") + 0.5
        dg, db, coef = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), torch.zeros(3, Cc, device="cuda")
        sums, tickets = N.stats_buffer(Cc), torch.zeros(N.VT_FIN_TICKETS, dtype=torch.int32, device="cuda")
        nb = sums.numel() * 8
        red = (vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, 1, N.VT_BF16, vp(sums))
        app = (vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(coef), vp(dz), Cc, M, Cc, 1, N.VT_BF16)

        def two(st):
            N.check(lib.vt_memset(sums.data_ptr(), 0, nb, st))
            N.check(lib.vt_bn_act_bwd_reduce(*red, st))
            N.check(lib.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 1.0, vp(scale), vp(mean), vp(invstd), 1, vp(dg), vp(db), vp(coef), st))
            N.check(lib.vt_bn_act_bwd_apply(*app, st))

        def one(st):
            N.check(lib.vt_memset(sums.data_ptr(), 0, nb, st))
            N.check(lib.vt_bn_act_bwd_reduce_finalize(*red, float(M), 1.0, 1, vp(dg), vp(db), vp(coef), vp(tickets), st))
            N.check(lib.vt_bn_act_bwd_apply(*app, st))

        def base(st):
            N.check(lib.vt_memset(sums.data_ptr(), 0, nb, st))
            N.check(lib.vt_bn_act_bwd_reduce(*red, st))
            N.check(lib.vt_bn_act_bwd_apply(*app, st))

        t2, t1, t0 = timeit(two), timeit(one), timeit(base)
        t2b, t1b, t0b = timeit(two), timeit(one), timeit(base)
        print(f"M {M:8d} C {Cc:5d}: memset+reduce+apply {min(t0, t0b):7.1f} us | + finalize launch {min(t2, t2b):7.1f} | "
              f"+ finalize tail {min(t1, t1b):7.1f}", flush=True)


main()
