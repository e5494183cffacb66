"""Helpers for the GPU parity tests: raw C-ABI calls on torch device tensors."""
from __future__ import annotations

import ctypes as C
import math

import torch

from vision_toolbox import _native as N

TD = {N.VT_F32: torch.float32, N.VT_BF16: torch.bfloat16}
DTYPES = [N.VT_F32, N.VT_BF16]
DNAME = {N.VT_F32: "f32", N.VT_BF16: "bf16"}


def stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def nhwc(t: torch.Tensor, dtype: int, ld: int | None = None, coff: int = 0) -> torch.Tensor:
    """NCHW cpu tensor -> device NHWC tensor (optionally a channel slice of a wider buffer)."""
    v = t.permute(0, 2, 3, 1).contiguous().to("cuda", TD[dtype])
    if ld is None:
        return v
    B, H, W, Cc = v.shape
    wide = torch.full((B, H, W, ld), float("nan"), device="cuda", dtype=TD[dtype])
    wide[..., coff : coff + Cc] = v
    return wide[..., coff : coff + Cc]


def to_nchw(v: torch.Tensor) -> torch.Tensor:
    return v.float().permute(0, 3, 1, 2).contiguous().cpu()


def krsc(w: torch.Tensor, dtype: int) -> torch.Tensor:
    """OIHW -> [O][kh][kw][I] on device."""
    return w.permute(0, 2, 3, 1).contiguous().to("cuda", TD[dtype])


def rounded(t: torch.Tensor, dtype: int) -> torch.Tensor:
    """what the kernel actually sees after storage in `dtype` (as fp32 on cpu)."""
    return t.to(TD[dtype]).float()


def conv_desc(dtype, x_nhwc, Cin, Cout, k, s, pad, ldy, flags=0, ldr=0) -> N.ConvDesc:
    B, H, W, _ = x_nhwc.shape
    d = N.ConvDesc()
    d.dtype = dtype
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, H, W, Cin, x_nhwc.stride(2)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = Ho, Wo, s, s, -pad, -pad
    d.Cout, d.ldy, d.oH, d.oW = Cout, ldy, Ho, Wo
    d.oHs = d.oWs = 1
    d.oh0 = d.ow0 = 0
    d.ldw, d.ldr, d.flags, d.ntaps = k * k * Cin, ldr, flags, k * k
    i = 0
    for r in range(k):
        for t in range(k):
            d.dh[i], d.dw[i] = r, t
            i += 1
    return d


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def tol(dtype: int, f32: float = 2e-5, bf16: float = 6e-3) -> float:
    return f32 if dtype == N.VT_F32 else bf16
