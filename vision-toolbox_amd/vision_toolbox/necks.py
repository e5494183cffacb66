"""FPN / PAN necks on the backbone kernels.

Same constructors, child names (`lateral_convs`, `output_convs`, `top_down`, `bottom_up`) and
therefore the same state_dict keys as the reference (vision_toolbox/necks.py:45-120); `forward`
takes the list a backbone's `get_feature_maps()` returns (bottom = largest map first) and returns a
list of the same length.  Execution: one compiled launch list per (shapes, dtype, mode) --
biased 1x1 lateral convs and 3x3 ConvNormAct output convs on the implicit-GEMM kernels, and the
`nn.Upsample(nearest, x2 | x0.5)` + `sum` pair as ONE resampling kernel (vt_resample2x_add_fwd);
the whole neck is one autograd.Function, so it composes with the backbone's.

Only what the reference's defaults use is implemented on the GPU: `fuse_fn="sum"`, nearest
interpolation, `ConvNormAct` blocks.  Other settings construct (state_dict parity) and raise
NotImplementedError when run; BiFPN (separable convs) is outside the hot path.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
from torch import Tensor, nn

from . import _native as N
from . import engine as E
from .components import ConvNormAct
from .program import Program, current_stream_handle, mode_signature, tracing_paused

__all__ = ["FPN", "PAN"]

_FUSE = ("concat", "sum", "avg", "max")


class _NeckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner: "_NeckRunner", prog: Program, n_in: int, *tensors):
        st, outs = runner._run_forward(prog, tensors[:n_in])
        ctx.runner, ctx.st, ctx.n_in, ctx.n_params = runner, st, n_in, len(tensors) - n_in
        ctx.in_meta = [(t.dtype, t.requires_grad) for t in tensors[:n_in]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        dxs, pgrads = ctx.runner._run_backward(ctx.st, gouts, ctx.in_meta)
        ctx.st = None
        return (None, None, None, *dxs, *pgrads[: ctx.n_params])


class _NeckRunner:
    def __init__(self, module: nn.Module):
        self.module = module
        self.store = E.ParamStore(module)
        self.cache: dict = {}

    def program(self, xs: Sequence[Tensor], dtype: int, need_grad: bool) -> Program:
        key = (tuple(tuple(x.shape) for x in xs), dtype, need_grad,
               tuple(bool(x.requires_grad) for x in xs), self.store.version, mode_signature(self.module, self.store))
        prog = self.cache.get(key)
        if prog is None:
            b = E.Builder(self.store, dtype, self.module.training, need_grad)
            if dtype == N.VT_BF16:
                n = self.store.pflat.numel()
                b.emit(N.OP_COPY2D, [(E.PARAMS, 0), (E.MIRROR, 0)], [N.VT_F32, N.VT_BF16, n, 0], [n, n, 1])
            refs = [b.input_map(x.shape[0], x.shape[1], x.shape[2], x.shape[3], f"in{i}",
                                requires_grad=bool(x.requires_grad) and need_grad) for i, x in enumerate(xs)]
            outs = self.module._vt_emit_list(b, refs)
            seeds = []
            if need_grad:
                seeds = b.seed_output_grads(outs)
                b.build_backward()
            prog = Program(b, outs, seeds)
            prog.ext_inputs, prog.ext_grads = list(b.ext_inputs), list(b.ext_grads)
            self.cache[key] = prog
        return prog

    def __call__(self, xs: Sequence[Tensor]):
        xs = list(xs)
        if not xs or any((not isinstance(x, Tensor)) or x.dim() != 4 for x in xs):
            raise ValueError("expected a list of 4-D NCHW feature maps")
        if all(not x.is_cuda for x in xs):
            return self.module._eager_list(xs)  # dispatch rule: CPU tensors -> plain torch ops
        if any(not x.is_cuda for x in xs):
            raise RuntimeError("feature maps on different devices")
        N.lib()  # a GPU tensor never falls back to the eager path
        if len({x.dtype for x in xs}) != 1:
            raise ValueError("feature maps of one dtype expected")
        if xs[0].dtype == torch.bfloat16 or (torch.is_autocast_enabled() and
                                              torch.get_autocast_gpu_dtype() == torch.bfloat16):
            dtype = N.VT_BF16
        elif xs[0].dtype == torch.float32:
            dtype = N.VT_F32
        else:
            raise TypeError(f"feature maps must be float32 or bfloat16, got {xs[0].dtype}")
        with tracing_paused():
            self.store.ensure(xs[0].device)
            params = self.store.params
            need_grad = torch.is_grad_enabled() and (any(x.requires_grad for x in xs) or
                                                     any(p.requires_grad for p in params))
            prog = self.program(xs, dtype, need_grad)
        if need_grad:
            outs = _NeckFn.apply(self, prog, len(xs), *xs, *params)
        else:
            _, outs = self._run_forward(prog, xs)
        return list(outs)

    def _run_forward(self, prog: Program, xs):
        dev = xs[0].device
        with torch.cuda.device(dev):
            arena = torch.empty(prog.arena_bytes, dtype=torch.uint8, device=dev)
            for ref, x in zip(prog.ext_inputs, xs):
                E.tref_to_tensor(arena, ref).copy_(x.detach())  # layout / dtype plumbing only
            s = self.store
            bases = prog.bases(arena.data_ptr(), PARAMS=s.pflat.data_ptr(), STATE=s.sflat.data_ptr(),
                               MIRROR=s.mirror.data_ptr(), COUNTERS=s.nflat.data_ptr())
            N.run_ops(prog.fwd_ops, prog.n_fwd, bases, current_stream_handle())
        outs = [E.tref_to_tensor(arena, t) for t in prog.outs]
        return (prog, arena, bases), outs

    def _run_backward(self, st, gouts, in_meta):
        prog, arena, bases = st
        if prog.n_bwd == 0:
            raise RuntimeError("this forward was compiled without a backward list (no-grad call)")
        with torch.cuda.device(arena.device):
            for seed, g in zip(prog.seeds, gouts):
                view = E.tref_to_tensor(arena, seed)
                if g is None:
                    view.zero_()
                else:
                    view.copy_(g)
            N.run_ops(prog.bwd_ops, prog.n_bwd, bases, current_stream_handle())
        dxs = []
        for gref, (dt, rg) in zip(prog.ext_grads, in_meta):
            if not rg or gref is None:
                dxs.append(None)
            else:
                dxs.append(E.tref_to_tensor(arena, gref).to(dt))
        zb = arena[prog.zb_off : prog.zb_off + prog.zb_bytes]
        pgrads = []
        for p in self.store.params:
            off = prog.param_grad_off.get(id(p))
            if off is None or not p.requires_grad:
                pgrads.append(None)
                continue
            g = zb[off : off + p.numel() * 4].view(torch.float32)
            if p.dim() == 4:
                o, i, kh, kw = p.shape
                g = g.view(o, kh, kw, i).permute(0, 3, 1, 2)
            else:
                g = g.view(p.shape)
            pgrads.append(g)
        return dxs, pgrads


class _NeckBase(nn.Module):
    _runner: Optional[_NeckRunner] = None

    def forward(self, x: "list[Tensor]") -> "list[Tensor]":
        r = self.__dict__.get("_runner")
        if r is None:
            r = _NeckRunner(self)
            self.__dict__["_runner"] = r
        return r(x)


# https://arxiv.org/abs/1612.03144
class FPN(_NeckBase):
    def __init__(
        self,
        in_channels_list: "list[int]",
        out_channels: int = 256,
        fuse_fn: str = "sum",
        block: Callable[[int, int], nn.Module] = ConvNormAct,
        interpolation_mode: str = "nearest",
        top_down: bool = True,
    ):
        super().__init__()
        if fuse_fn not in _FUSE:
            raise KeyError(fuse_fn)
        self.fuse_fn = fuse_fn
        self.interpolation_mode = interpolation_mode
        self.out_channels = out_channels
        self.top_down = top_down
        self.lateral_convs = nn.ModuleList(
            [nn.Conv2d(in_c, out_channels, kernel_size=1) if in_c != out_channels else nn.Identity()
             for in_c in in_channels_list]
        )
        self.upsample = nn.Upsample(scale_factor=2.0 if top_down else 0.5, mode=interpolation_mode)
        in_c = out_channels if fuse_fn == "sum" else out_channels * 2
        self.output_convs = nn.ModuleList([block(in_c, out_channels) for _ in range(len(in_channels_list) - 1)])

    def _vt_emit_list(self, b, xs, name: str = "fpn"):
        if self.fuse_fn not in ("sum", "concat") or self.interpolation_mode not in ("nearest", "bilinear"):
            # ('avg' / 'max' cannot run in the reference either: its output convs are built for 2 x out_channels whenever
            #  fuse_fn != 'sum', necks.py:66, while those two fuse functions return out_channels)
            raise NotImplementedError("the MI355X neck implements fuse_fn 'sum' / 'concat' with nearest / bilinear resampling")
        assert len(xs) == len(self.lateral_convs)
        outs = []
        for i, (lat, x) in enumerate(zip(self.lateral_convs, xs)):
            if isinstance(lat, nn.Identity):
                outs.append(x)
            else:
                outs.append(b.conv_unit(x, lat, None, False, name=f"{name}.lateral_convs.{i}"))
        n = len(outs)
        for i, oc in enumerate(self.output_convs):
            if not isinstance(oc, ConvNormAct):
                raise NotImplementedError("neck blocks other than ConvNormAct are outside the hot path")
            # top-down: levels n-2, ..., 0 receive the level above, upsampled (necks.py:70-73); bottom-up: levels 1, ..., n-1
            # the level below, subsampled (necks.py:76-79)
            dst, src = (n - 2 - i, n - 1 - i) if self.top_down else (i + 1, i)
            mode = (0 if self.top_down else 1) + (2 if self.interpolation_mode == "bilinear" else 0)
            if self.fuse_fn == "sum":
                fused = b.resample_add(outs[src], outs[dst], mode, name=f"{name}.fuse.{dst}")
            else:  # torch.cat([x_dst, resample(x_src)], dim=1) (necks.py:14-15): two channel slices of one buffer
                d = outs[dst]
                fused = b.act(d.B, d.H, d.W, 2 * d.C, f"{name}.cat.{dst}")
                b.copy(d, fused.sl(0, d.C))
                b.resample_add(outs[src], None, mode, name=f"{name}.fuse.{dst}", out=fused.sl(d.C, d.C))
            outs[dst] = oc._vt_emit(b, fused, name=f"{name}.output_convs.{i}")
        return outs

    def _eager_list(self, xs):
        """CPU tensors: the same wiring with torch ops over this module's children."""
        if self.fuse_fn not in ("sum", "concat"):
            raise NotImplementedError("fuse_fn 'sum' / 'concat' (the reference's 'avg' / 'max' cannot run there either)")
        assert len(xs) == len(self.lateral_convs)
        outs = [lat(x) for lat, x in zip(self.lateral_convs, xs)]
        n = len(outs)
        for i, oc in enumerate(self.output_convs):
            dst, src = (n - 2 - i, n - 1 - i) if self.top_down else (i + 1, i)
            up = self.upsample(outs[src])
            outs[dst] = oc(outs[dst] + up if self.fuse_fn == "sum" else torch.cat([outs[dst], up], dim=1))
        return outs


# https://arxiv.org/abs/1803.01534
class PAN(_NeckBase):
    def __init__(
        self,
        in_channels_list: "list[int]",
        out_channels: int = 256,
        fuse_fn: str = "sum",
        block: Callable[[int, int], nn.Module] = ConvNormAct,
        interpolation_mode: str = "nearest",
    ):
        super().__init__()
        self.top_down = FPN(in_channels_list, out_channels, fuse_fn=fuse_fn, block=block,
                            interpolation_mode=interpolation_mode)
        self.bottom_up = FPN([out_channels] * len(in_channels_list), out_channels, fuse_fn=fuse_fn, block=block,
                             interpolation_mode=interpolation_mode)

    def _vt_emit_list(self, b, xs, name: str = "pan"):
        return self.bottom_up._vt_emit_list(b, self.top_down._vt_emit_list(b, xs, name + ".top_down"),
                                            name + ".bottom_up")

    def _eager_list(self, xs):
        return self.bottom_up._eager_list(self.top_down._eager_list(xs))
