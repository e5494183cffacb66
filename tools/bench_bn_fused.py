"""Isolated timing (GPU box): vt_bn_act_bwd_fused against the three launches it replaces, CSPDarknet-53 shapes at batch 256.

    python tools/bench_bn_fused.py"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N


def timeit(fn, iters=30, warmup=8):
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
    for M, Cc in [(256 * 14 * 14, 256), (256 * 7 * 7, 512), (256 * 7 * 7, 1024), (128 * 14 * 14, 256), (256 * 28 * 28, 128)]:
        dy = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
        z = (torch.randn(M, Cc, device="cuda") + 0.3).to(torch.bfloat16)
        dz = torch.empty_like(dy)
        scale, shift = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
        mean, invstd = torch.randn(Cc, device="cuda") * 0.1 + 0.3, torch.rand(Cc, device="cuda") + 0.5
        dg, db, coef = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), torch.empty(3, Cc, device="cuda")
        sums, sync = N.stats_buffer(Cc), torch.zeros(4, dtype=torch.int32, device="cuda")

        def three(st):
            sums.zero_()
            N.check(lib.vt_bn_act_bwd_reduce(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, 1, N.VT_BF16, vp(sums), st))
            N.check(lib.vt_bn_bwd_finalize(vp(sums), Cc, float(M), 1.0, vp(scale), vp(mean), vp(invstd), 1, vp(dg), vp(db), vp(coef), st))
            N.check(lib.vt_bn_act_bwd_apply(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(coef), vp(dz), Cc, M, Cc, 1, N.VT_BF16, st))

        def one(st):
            sums.zero_()
            sync.zero_()
            N.check(lib.vt_bn_act_bwd_fused(vp(dy), Cc, vp(z), Cc, vp(scale), vp(shift), vp(mean), vp(invstd), M, Cc, 1, N.VT_BF16, float(M), 1.0,
                                            1, vp(sums), vp(sync), vp(dg), vp(db), vp(coef), vp(dz), Cc, st))

        def zero_only(st):
            sums.zero_()
            sync.zero_()

        t3, t1, t0 = timeit(three), timeit(one), timeit(zero_only)
        name = N.last_kernel_name()
        print(f"M {M:7d} C {Cc:5d}: three launches {t3 - t0 / 2:7.1f} us | one launch {t1 - t0:7.1f} us [{name}] (two zeroing launches {t0:.1f} us subtracted)", flush=True)
    n = C.c_uint32(0)
    N.check(lib.vt_bn_bwd_fused_timeouts(C.byref(n)))
    print("barrier timeouts:", n.value)


if __name__ == "__main__":
    main()
