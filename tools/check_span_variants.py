"""Dev check (GPU box): dispatcher variants of the 3x3 span kernels (default: span6 against the span kernel) on the
same operands -- outputs, BN statistics -- and interleaved timing rounds.   python tools/check_span_variants.py [Cin,Cout,H[,B]] ..."""
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


# (tag, environment) per variant; the first one is the reference.  VT_CHECK_VARIANTS="tag:K=V,K=V;tag2:K=V" overrides
VARIANTS = [("span", {"VT_SPAN6": "0"}), ("span6", {"VT_SPAN6": "2"})]
if os.environ.get("VT_CHECK_VARIANTS"):
    VARIANTS = [(v.split(":")[0], dict(kv.split("=") for kv in v.split(":")[1].split(",") if kv))
                for v in os.environ["VT_CHECK_VARIANTS"].split(";")]


def set_knobs(allkeys, env):
    """the dispatchers read the environment once per process: variants are switched through vt_set_knob"""
    for k in allkeys:
        N.set_knob(k, int(env.get(k, "0")))


ROUNDS = int(os.environ.get("VT_CHECK_ROUNDS", "7"))


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
        allkeys = {k for _, env in VARIANTS for k in env}
        for mode, flags in (("stats", N.VT_CONV_STATS), ("residual", N.VT_CONV_RESIDUAL)):
            outs = {}
            d = desc(B, Cin, Cout, H, flags, ldr=Cout if flags & N.VT_CONV_RESIDUAL else 0)
            r_ = res if flags & N.VT_CONV_RESIDUAL else None
            for tag, env in VARIANTS:
                set_knobs(allkeys, env)
                y = torch.full((B, H, H, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
                st = N.stats_buffer(Cout) if flags & N.VT_CONV_STATS else None
                name, _ = run(d, x, w, y, st, r_)
                outs[tag] = [y, st.double().sum(0).cpu() if st is not None else None, name, []]
            if mode == "stats":  # interleaved timing rounds (guide rule 24): median and min per variant
                st = N.stats_buffer(Cout)
                ysc = torch.empty_like(outs[VARIANTS[0][0]][0])
                for _ in range(ROUNDS):
                    for tag, env in VARIANTS:
                        set_knobs(allkeys, env)
                        outs[tag][3].append(run(d, x, w, ysc, st, r_, iters=10)[1])
            y0, s0 = outs[VARIANTS[0][0]][0], outs[VARIANTS[0][0]][1]
            line = f"{Cin}->{Cout} @{H} B={B} {mode:8s}"
            for tag in [v[0] for v in VARIANTS]:
                y, ssum, name, mss = outs[tag]
                nan = int(torch.isnan(y.float()).sum())
                neq = int((y != y0).sum())
                serr = ((ssum - s0).abs().max() / s0.abs().max()).item() if ssum is not None else 0.0
                line += f" | {tag} [{name.replace('span_kernel', 'sk')}] neq {neq} nan {nan} st {serr:.1g}"
                if mss:
                    mss = sorted(mss)
                    med = mss[len(mss) // 2]
                    line += f" med {med*1e3:.1f}us {flops/med/1e9:.0f}TF min {mss[0]*1e3:.1f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
