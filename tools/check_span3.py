"""Dev check (GPU box): the persistent span kernel (vt_igemm_span3.hip) against the shipped span kernel on the
same operands -- outputs, BN statistics -- and both timed.   python tools/check_span3.py [Cin,Cout,H[,B]] ..."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch

from vision_toolbox import _native as N

lib = N.lib()


def desc(B, Cin, Cout, H, flags, ldy=None, ldr=0):
    d = N.ConvDesc()
    d.dtype = N.VT_BF16
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, H, Cin, Cin
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = H, H, 1, 1, -1, -1
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, ldy or Cout, H, H, 1, 1
    d.ldw, d.flags, d.ntaps, d.ldr = 9 * Cin, flags, 9, ldr
    for i in range(9):
        d.dh[i], d.dw[i] = i // 3, i % 3
    return d


def run(d, x, w, y, stats, res=None, iters=0):
    s = int(torch.cuda.current_stream().cuda_stream)
    f = lambda: N.check(lib.vt_conv_igemm(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None,
                                          res.data_ptr() if res is not None else None,
                                          stats.data_ptr() if stats is not None else None, s))
    f()
    name = N.last_kernel_name()
    ms = 0.0
    if iters:
        for _ in range(5):
            f()
        e0, e1 = N.Event(), N.Event()
        e0.record(s)
        for _ in range(iters):
            f()
        e1.record(s)
        ms = e0.elapsed_ms(e1) / iters
    torch.cuda.synchronize()
    return name, ms


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(128, 128, 28), (256, 256, 14), (160, 160, 28),
                                                                            (128, 128, 56), (64, 64, 56), (512, 512, 7)]
    for sh in shapes:
        Cin, Cout, H = sh[:3]
        B = sh[3] if len(sh) > 3 else 256
        torch.manual_seed(0)
        x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
        res = torch.randn(B, H, H, Cout, device="cuda").to(torch.bfloat16)
        flops = 2.0 * B * H * H * Cout * 9 * Cin
        for mode, flags in (("stats", N.VT_CONV_STATS), ("residual", N.VT_CONV_RESIDUAL)):
            outs = {}
            for tag, env in (("old", {"VT_SPAN3": "0"}), ("wm4", {"VT_SPAN3": "1", "VT_SPAN3_WM": "4"}),
                             ("wm2", {"VT_SPAN3": "1", "VT_SPAN3_WM": "2"})):
                os.environ.update(env)
                y = torch.full((B, H, H, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
                st = torch.zeros(N.VT_STAT_REPLICAS, 2, Cout, device="cuda") if flags & N.VT_CONV_STATS else None
                d = desc(B, Cin, Cout, H, flags, ldr=Cout if flags & N.VT_CONV_RESIDUAL else 0)
                name, _ = run(d, x, w, y, st, res if flags & N.VT_CONV_RESIDUAL else None)
                st_sum = st.double().sum(0).cpu() if st is not None else None
                _, ms = run(d, x, w, y, torch.zeros_like(st) if st is not None else None,
                            res if flags & N.VT_CONV_RESIDUAL else None, iters=20 if mode == "stats" else 0)
                outs[tag] = (y, st_sum, name, ms)
            y0, s0 = outs["old"][0], outs["old"][1]
            line = f"{Cin}->{Cout} @{H} B={B} {mode:8s}"
            for tag in ("old", "wm4", "wm2"):
                y, ssum, name, ms = outs[tag]
                nan = int(torch.isnan(y.float()).sum())
                diff = (y.float() - y0.float()).abs().max().item() if nan == 0 else float("nan")
                neq = int((y != y0).sum())
                serr = ((ssum - s0).abs().max() / s0.abs().max()).item() if ssum is not None else 0.0
                line += f" | {tag} [{name}] maxdiff {diff:.3g} neq {neq} nan {nan} stats {serr:.2g}"
                if ms:
                    line += f" {ms*1e3:.1f}us {flops/ms/1e9:.0f}TF"
            print(line, flush=True)


if __name__ == "__main__":
    main()
