"""Harness of the consumer-side-normalise prototype in span6's loader waves (tools/diag/libvt_s6proto.so, built with
-DVT_SPAN6_PROTO_NORM).  The prototype launch computes conv(relu(x)):

    VT_AMD_LIB=tools/diag/libvt_s6proto.so python tools/s6proto_check.py proto <Cin,Cout,H> ...   (x with negative values)
    python tools/s6proto_check.py ref <Cin,Cout,H> ...                                           (the shipped kernel on relu(x))
    python tools/s6proto_check.py compare <Cin,Cout,H> ...

Each run prints the time per launch (statistics epilogue, batch 256) and stores its output; `compare` wants them bit-equal."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT), str(ROOT / "tools")]

import torch

from vision_toolbox import _native as N
from bench_conv import desc_for, timeit

OUT = Path("/tmp/s6proto")
OUT.mkdir(exist_ok=True)


def main():
    mode = sys.argv[1]
    for spec in sys.argv[2:]:
        Cin, Cout, H = (int(v) for v in spec.split(","))
        f = OUT / f"s6proto_{Cin}_{Cout}_{H}"
        if mode == "compare":
            a, b = torch.load(str(f) + "_ref.pt"), torch.load(str(f) + "_proto.pt")
            print(spec, "bit-equal" if torch.equal(a, b) else f"DIFFER max {float((a.float() - b.float()).abs().max()):.3e}")
            continue
        B = 256
        g = torch.Generator(device="cuda").manual_seed(7)
        x = torch.randn(B, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
        if mode == "ref":
            x = torch.relu(x)
        w = (torch.randn(Cout, 9, Cin, device="cuda", generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
        d, Ho = desc_for(B, Cin, Cout, 3, 1, H, N.VT_CONV_STATS)
        y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=torch.bfloat16)
        stats = N.stats_buffer(Cout)
        lib = N.lib()
        ms = timeit(lambda st: N.check(lib.vt_conv_igemm(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None,
                                                         None, stats.data_ptr(), st)))
        torch.cuda.synchronize()
        print(f"{mode} {spec}: {ms * 1e3:.1f} us per launch [{N.last_kernel_name()}]", flush=True)
        torch.save(y.cpu(), str(f) + f"_{mode}.pt")


if __name__ == "__main__":
    main()
