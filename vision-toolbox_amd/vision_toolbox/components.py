"""ConvNormAct -- the fused Conv -> BatchNorm -> ReLU unit of the hot path.

Same constructor signature, child names (`conv`, `norm`, `act`) and therefore the same
state_dict keys as the reference unit (vision_toolbox/components.py:13-46), and the
children are real nn.Conv2d / nn.BatchNorm2d instances so that the reference's
isinstance-based parameter grouping (classifier.py:111-155) keeps working.  What differs
is execution: on the GPU the unit is not three ATen calls but a few launches of
libvt_amd (implicit-GEMM conv on MFMA with the BN statistics in its epilogue, one
normalise+ReLU(+residual) pass), emitted into a static launch list by `_vt_emit`.
CPU tensors run the children with plain torch ops (`_eager_maps`), as in the reference.

Every constructor argument of the reference unit runs on the GPU: the Darknet / VoVNet configuration (bn + relu, groups
= dilation = 1) on the fused kernels; other activations, `groups` (>= one 16-byte channel chunk per group), `dilation`
and norm="none" on the general kernels (engine.Builder.conv_unit / _grouped_unit).  Depthwise units (groups = in_channels = out_channels) run on the streaming kernels of vt_dwconv.hip (round 6); other groups narrower than a 16-byte channel slice raise.
"""
from __future__ import annotations

import math
import weakref
from typing import Optional

import torch
from torch import Tensor, nn

__all__ = ["ConvNormAct", "HipModule"]

_RUNNERS: "weakref.WeakKeyDictionary[nn.Module, object]" = weakref.WeakKeyDictionary()


class HipModule(nn.Module):
    """Mixin: a module whose forward is a compiled libvt_amd program.

    Subclasses implement `_vt_emit_maps(builder, x_ref) -> list[TRef]`.
    `compute_dtype` (None / torch.float32 / torch.bfloat16) overrides the dtype rule
    (f32 input -> exact-f32 kernels, bf16 input or bf16 autocast -> bf16 kernels).
    """

    compute_dtype: Optional[torch.dtype] = None

    def _vt_runner(self):
        r = _RUNNERS.get(self)
        if r is None:
            from .program import BackboneRunner

            r = BackboneRunner(self)
            _RUNNERS[self] = r
        return r

    def _vt_emit_maps(self, b, x):  # pragma: no cover - abstract
        raise NotImplementedError

    def _vt_emit(self, b, x, out=None):
        return self._vt_emit_maps(b, x)[-1]

    # Dispatch rule: the SAME module class serves CPU tensors through plain torch ops over
    # its own nn.Conv2d / nn.BatchNorm2d / nn.ReLU children (`_eager*`), and GPU tensors through
    # libvt_amd.  The runner picks by `x.is_cuda`; a GPU tensor never takes the eager path.
    def _eager_maps(self, x: Tensor) -> "list[Tensor]":  # pragma: no cover - abstract
        raise NotImplementedError

    def _eager(self, x: Tensor) -> Tensor:
        return self._eager_maps(x)[-1]

    def forward(self, x: Tensor) -> Tensor:
        return self._vt_runner()(x, all_maps=False, compute_dtype=self.compute_dtype)[-1]


_ACTS = {
    "none": lambda: nn.Identity(),
    "relu": lambda: nn.ReLU(inplace=True),
    "leaky_relu": lambda: nn.LeakyReLU(0.2, inplace=True),
    "swish": lambda: nn.SiLU(inplace=True),
    "silu": lambda: nn.SiLU(inplace=True),
    "gelu": lambda: nn.GELU(),
}
_NORMS = {"none": lambda c: nn.Identity(), "bn": lambda c: nn.BatchNorm2d(c)}


class ConvNormAct(nn.Sequential, HipModule):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: int = 3,
        stride: int = 1,
        dilation: int = 1,
        groups: int = 1,
        norm: str = "bn",
        act: str = "relu",
    ):
        if norm not in _NORMS:
            raise KeyError(norm)
        if act not in _ACTS:
            raise KeyError(act)
        super().__init__()
        # "same"-style padding of the reference: ceil((k - s) / 2)  (components.py:31)
        pad = -((stride - kernel_size) // 2)
        assert pad == math.ceil((kernel_size - stride) / 2)
        self.add_module(
            "conv",
            nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, dilation=dilation,
                      groups=groups, bias=(norm == "none")),
        )
        self.add_module("norm", _NORMS[norm](out_channels))
        self.add_module("act", _ACTS[act]())
        self._act_name = act
        if act in ("relu", "leaky_relu"):
            # N(0, sqrt(2 / ((1 + 0.2^2) * fan_out)))  (components.py:45-46)
            nn.init.kaiming_normal_(self.conv.weight, a=0.2, mode="fan_out", nonlinearity=act)

    # -- launch-list emission ----------------------------------------------------------
    def _vt_relu(self) -> int:
        """activation code of the kernels (include/vt_amd.h): 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 SiLU, 4 GELU (exact)"""
        a = self.act
        if isinstance(a, nn.Identity):
            return 0
        if isinstance(a, nn.ReLU):
            return 1
        if isinstance(a, nn.LeakyReLU) and abs(a.negative_slope - 0.2) < 1e-12:
            return 2
        if isinstance(a, nn.SiLU):
            return 3
        if isinstance(a, nn.GELU) and getattr(a, "approximate", "none") == "none":
            return 4
        raise NotImplementedError(
            f"activation {a!r}: the MI355X path implements ConvNormAct's own choices (components.py:37-44: none, relu, "
            "leaky_relu(0.2), swish / silu, gelu)"
        )

    def _vt_emit(self, b, x, out=None, residual=None, name: str = "cna", pool_out=None):
        """`pool_out`: also write MaxPool2d(3, 2, 1) of the unit's output there (VoVNet's `stage.max_pool`, vovnet.py:94):
        fused into the unit's normalise pass where one exists, else a pool launch"""
        norm = self.norm if isinstance(self.norm, nn.BatchNorm2d) else None
        return b.conv_unit(x, self.conv, norm, self._vt_relu(), residual=residual, out=out, name=name, pool_out=pool_out)

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x: Tensor) -> "list[Tensor]":
        return [self.act(self.norm(self.conv(x)))]

    def forward(self, x: Tensor) -> Tensor:
        return HipModule.forward(self, x)
