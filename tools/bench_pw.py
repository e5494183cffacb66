"""Dev bench (GPU box): one pointwise kernel (vt_pointwise.hip) on a layer shape, HIP events on the launch stream.

    python tools/bench_pw.py <mode> <K> <C0>[+<C1>] <M> [iters]      mode: stats | apply | reduce | bwd

Prints ms and the TB/s of the pass's operands (each once).  Under rocprofv3 --pmc it is the target of the counter passes.
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N

BF = torch.bfloat16


def arr(ct, vals):
    return (ct * len(vals))(*vals)


def vps(ts):
    return arr(C.c_void_p, [C.c_void_p(t.data_ptr()) if t is not None else None for t in ts])


def main():
    mode, K, cs, M = sys.argv[1], int(sys.argv[2]), [int(v) for v in sys.argv[3].split("+")], int(sys.argv[4])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    Nn = sum(cs)
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(c, K, device="cuda") * (2.0 / K) ** 0.5).to(BF) for c in cs]
    d = N.PwDesc()
    d.dtype, d.K, d.ngroups, d.relu, d.M = N.VT_BF16, K, len(cs), 1, M
    d.x, d.ldx = x.data_ptr(), K
    for g, w in enumerate(ws):
        d.C[g], d.w[g], d.ldw[g] = w.shape[0], w.data_ptr(), K
    coef = torch.rand(4, Nn, device="cuda") + 0.5
    ys = [torch.randn(M, c, device="cuda").to(BF) for c in cs]
    ld = arr(C.c_int32, cs)
    stats = [N.stats_buffer(c) for c in cs]
    bcoef = [torch.rand(3, c, device="cuda") * 0.1 for c in cs]
    dx = torch.zeros(M, K, device="cuda", dtype=BF)
    full = N.lib().vt_pw_supported(N.VT_BF16, K, cs[0], cs[1] if len(cs) > 1 else 0) == 2
    dws = [torch.zeros(c, K, device="cuda") if full else None for c in cs]
    dzs = [None if full else torch.zeros(M, c, device="cuda", dtype=BF) for c in cs]
    import os
    ress = [torch.randn(M, c, device="cuda").to(BF) if os.environ.get("VT_BENCH_RESIDUAL") else None for c in cs]  # apply + residual
    s = int(torch.cuda.current_stream().cuda_stream)
    lib = N.lib()

    def launch():
        if mode == "stats":
            N.check(lib.vt_pw_fwd_stats(C.byref(d), vps(stats), s))
        elif mode == "apply":
            N.check(lib.vt_pw_fwd_apply(C.byref(d), coef.data_ptr(), vps(ys), ld, vps(ress), ld if ress[0] is not None else arr(C.c_int32, [0, 0]), s))
        elif mode == "reduce":
            N.check(lib.vt_pw_bwd_reduce(C.byref(d), coef.data_ptr(), vps(ys), ld, vps(stats), s))
        else:
            N.check(lib.vt_pw_bwd_apply(C.byref(d), coef.data_ptr(), vps(ys), ld, vps(bcoef), dx.data_ptr(), K, dx.data_ptr(), K,
                                        vps(dws), arr(C.c_int32, [K] * len(cs)), vps(dzs), ld, s))

    for _ in range(3):
        launch()
    e0, e1 = N.Event(), N.Event()
    e0.record(s)
    for _ in range(iters):
        launch()
    e1.record(s)
    ms = e0.elapsed_ms(e1) / iters
    nb = 2.0 * M * K + {"stats": 0, "apply": 2.0 * M * Nn, "reduce": 2.0 * M * Nn,
                        "bwd": 2.0 * M * Nn + 4.0 * M * K + (0 if full else 2.0 * M * Nn)}[mode]
    print(f"{mode} {K}->{sys.argv[3]} M={M}: {ms * 1e3:.1f} us  {nb / ms / 1e9:.2f} TB/s  [{N.last_kernel_name()}]", flush=True)


if __name__ == "__main__":
    main()
