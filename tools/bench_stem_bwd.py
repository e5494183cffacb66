"""Time vt_stem_bn_bwd_reduce (vt_stem_bwd.hip) on the real stem geometry (GPU box):
    python tools/bench_stem_bwd.py [B=256] [H=224]
Prints ms per launch and the HBM rate of its algorithmic bytes (dy + z + x read once)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import ctypes as C

import torch

from vision_toolbox import _native as N


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    lib = N.lib()
    dev = "cuda"
    x = torch.randn(B, H, H, 8, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, 32, device=dev).to(torch.bfloat16)
    z = torch.randn(B, H, H, 32, device=dev).to(torch.bfloat16)
    co = [torch.randn(32, device=dev) for _ in range(4)]
    sums = N.stats_buffer(32)
    gzx = torch.zeros(lib.vt_stem_bn_bwd_scratch_bytes(32) // 4, device=dev)
    st = int(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())

    def run():
        N.check(lib.vt_stem_bn_bwd_reduce(N.VT_BF16, B, H, H, 32, vp(x), vp(dy), 32, vp(z), 32, vp(co[0]), vp(co[1]),
                                          vp(co[2]), vp(co[3]), 1, vp(sums), vp(gzx), int(os.environ.get('FIXED', '0')), st))
    for _ in range(3):
        run()
    e0, e1 = N.Event(), N.Event()
    e0.record(st)
    for _ in range(10):
        run()
    e1.record(st)
    ms = e0.elapsed_ms(e1) / 10
    nbytes = B * H * H * (8 + 32 + 32) * 2
    print(f"stem_bwd B={B} {H}x{H}: {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s of {nbytes / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
