"""Diagnostic: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the 1x1 concat convolutions of VoVNet-39
(vovnet.py:50-63) at batch 256 -- a yardstick for the hand-written kernels, not part of the product."""
import torch

SHAPES = [(802816, 768, 256), (200704, 1056, 512), (50176, 1472, 768), (12544, 1888, 1024), (802816, 128, 128), (200704, 256, 256)]


def timeit(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for M, K, N in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    dz = torch.randn(M, N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    gf = 2.0 * M * K * N / 1e9
    t_f = timeit(lambda: torch.mm(x, w.t(), out=y))
    t_d = timeit(lambda: torch.mm(dz, w, out=dx))
    gb = (M * K + M * N) * 2 / 1e9
    print(f"M={M} K={K} N={N} {gf:7.1f} GF | fwd {t_f:.4f} ms {gf / t_f:7.1f} TF/s {gb / t_f:5.2f} TB/s | dgrad {t_d:.4f} ms {gf / t_d:7.1f} TF/s", flush=True)
