"""Compiled programs and the autograd bridge of the module API.

`BackboneRunner` owns, per backbone instance, the flat parameter store and a cache of
compiled programs keyed by (input shape, dtype, mode).  `forward()` /
`get_feature_maps()` of a backbone (reference backbones/base.py:16-21) go through ONE
registered operator, `torch.ops.vision_toolbox_amd.backbone` (torch.library custom op,
SURVEY 8b): its CUDA implementation runs the forward launch list, its registered
autograd formula the explicit backward list, and its fake-tensor (meta) implementation
gives shapes / strides / dtypes without touching the GPU -- so `torch.jit.trace`
(reference tests/test_backbones.py:76-78) records a real, serialisable operator node and
`torch.compile` / `torch.export` can trace through the module in INFERENCE mode
(need_grad=False).  Not supported: tracing a training forward under fake tensors
(AOTAutograd / `torch.compile` of a training step) -- the run state of a forward lives
in this process, keyed by a token the fake implementation cannot produce, so the traced
backward raises; and loading a saved trace in ANOTHER process (the graph holds this
process's `handle`).  Eager autograd is the training path.  The returned feature maps are
ordinary autograd-tracked tensors, so user heads / necks (necks.py:83) compose with them.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import itertools
import weakref
from typing import Optional, Sequence

import torch
from torch import nn

from . import _native as N
from . import engine as E


@contextlib.contextmanager
def tracing_paused():
    """host-side set-up (flat parameter store, program compilation) must not be recorded by
    torch.jit.trace; only the `vision_toolbox_amd::backbone` operator call is."""
    state = torch._C._get_tracing_state()
    if state is None:
        yield
        return
    torch._C._set_tracing_state(None)
    try:
        yield
    finally:
        torch._C._set_tracing_state(state)


def current_stream_handle() -> int:
    # the raw hipStream_t of the current torch stream (also valid while dynamo runs the operator from a compiled
    # region, where torch.cuda.current_stream() returns a proxy without .cuda_stream)
    return int(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def resolve_dtype(x: torch.Tensor, override: Optional[torch.dtype]) -> int:
    """f32 input -> exact-f32 kernels; bf16 input or bf16 autocast -> bf16 kernels."""
    if override is not None:
        dt = override
    elif torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
        dt = torch.bfloat16
    else:
        dt = x.dtype
    if dt == torch.float32:
        return N.VT_F32
    if dt == torch.bfloat16:
        return N.VT_BF16
    raise TypeError(f"vision_toolbox (MI355X): compute dtype {dt} is not supported (float32 or bfloat16)")


def mode_signature(module: nn.Module, store: "E.ParamStore") -> tuple:
    """what a compiled program bakes in besides shapes: the train/eval flag of every BatchNorm (a
    frozen `bn.eval()` inside a training model must use its running statistics, as nn.BatchNorm2d
    does) and which parameters receive gradients."""
    bn = tuple(m.training for m in module.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))
    return bn, tuple(p.requires_grad for p in store.params)


class Program:
    """Forward (+ backward) launch lists over one arena."""

    def __init__(self, b: E.Builder, outs: Sequence[E.TRef], seeds: Sequence[E.TRef]):
        self.outs = list(outs)
        self.seeds = list(seeds)
        self.dtype = b.dtype
        top = E._round_up(b.arena_top, E.ALIGN)
        self.zf_off, self.zf_bytes = top, b.zf_top
        top = E._round_up(top + b.zf_top, E.ALIGN)
        self.zb_off, self.zb_bytes = top, b.zb_top
        self.arena_bytes = E._round_up(top + b.zb_top, E.ALIGN) + E.ALIGN
        fwd, bwd = list(b.fwd), list(b.bwd)
        # eval-mode BatchNorm coefficients depend on parameters and buffers only: all of them go to the head of the list,
        # where the executor turns the run into one launch per 40 BatchNorms (73 launches of ~5 us in a YOLOv5x forward)
        coefs = [op for op in fwd if (op.kind & 0xFFFF) == N.OP_BN_EVAL_COEFFS and not (op.kind & N.OP_SIDE_STREAM)]
        if len(coefs) > 1:
            ids = {id(op) for op in coefs}
            fwd = coefs + [op for op in fwd if id(op) not in ids]
        if self.zf_bytes:
            fwd.insert(0, _memset_op(E.ZERO_F, self.zf_bytes))
        if self.zb_bytes and bwd:
            bwd.insert(0, _memset_op(E.ZERO_B, self.zb_bytes))
        self.n_fwd, self.n_bwd = len(fwd), len(bwd)
        self.fwd_ops = E.ops_array(fwd)
        self.bwd_ops = E.ops_array(bwd)
        self.param_grad_off = dict(b.param_grad_off)
        self.input_grad = getattr(b, "input_grad", None)
        self.n_units = b.n_units
        self.builder = b  # kept for tools/debug_*.py (name -> activation / gradient buffers)
        self.kind_histogram = {}
        for op in fwd + bwd:
            k = N.OP_NAMES.get(op.kind & 0xFFFF, str(op.kind))
            self.kind_histogram[k] = self.kind_histogram.get(k, 0) + 1

    def bases(self, arena_ptr: int, **named) -> list:
        bs = [None] * E.NUM_BASES
        bs[E.ARENA] = arena_ptr
        bs[E.ZERO_F] = arena_ptr + self.zf_off
        bs[E.ZERO_B] = arena_ptr + self.zb_off
        for k, v in named.items():
            bs[getattr(E, k)] = v
        return bs


def _memset_op(base: int, nbytes: int) -> N.Op:
    op = N.Op()
    op.kind = N.OP_MEMSET
    for k in range(N.VT_OP_MAX_PTR):
        op.ptr[k].base = -1
    op.ptr[0].base = base
    op.ptr[0].offset = 0
    op.i[0] = 0
    op.f[0] = float(nbytes)
    return op


class _RunState:
    __slots__ = ("prog", "arena", "bases", "x_shape")


# ---- the registered operator ---------------------------------------------------------------------
# Non-tensor state travels by handle: `handle` names the BackboneRunner (weak registry); the run state a
# backward needs (arena, bases) is parked by the CUDA implementation under `token` (its last, CPU int64,
# output) and moved into the autograd context by setup_context, so it lives exactly as long as the graph.
_RUNNERS: "weakref.WeakValueDictionary[int, BackboneRunner]" = weakref.WeakValueDictionary()
_HANDLES = itertools.count(1)
_TOKENS = itertools.count(1)
_TORCH_OUT_DTYPE = {N.VT_F32: torch.float32, N.VT_BF16: torch.bfloat16}


def _runner(handle: int) -> "BackboneRunner":
    r = _RUNNERS.get(handle)
    if r is None:
        raise RuntimeError("vision_toolbox_amd::backbone: the backbone this graph was traced with no longer exists")
    return r


# (defined through torch.library.define / impl rather than the custom_op decorator: the feature maps are views of
#  ONE arena allocation, which the decorator's output-aliasing check rejects; they never alias an input)
torch.library.define("vision_toolbox_amd::backbone",
                     "(Tensor x, Tensor[] params, int handle, bool all_maps, int dtype, bool need_grad) -> Tensor[]")


@torch.library.impl("vision_toolbox_amd::backbone", "CUDA")
def _backbone_cuda(x: torch.Tensor, params: list[torch.Tensor], handle: int, all_maps: bool, dtype: int,
                   need_grad: bool) -> list[torch.Tensor]:
    """feature maps (logical NCHW, channels_last strides) of backbone `handle` for images x; `params` are the
    module's parameters in flat-store order (read from the store; listed so that autograd routes gradients)."""
    runner = _runner(handle)
    with tracing_paused():
        prog = runner.program(x, dtype, all_maps, need_grad)
        st, outs = runner._run_forward(prog, x)
    token = next(_TOKENS)
    if need_grad:
        runner._pending[token] = st
        while len(runner._pending) > 4:  # forwards whose graph was never built: do not hoard their arenas
            runner._pending.pop(next(iter(runner._pending)))
    return list(outs) + [torch.tensor([token], dtype=torch.int64)]


@torch.library.register_fake("vision_toolbox_amd::backbone")
def _backbone_fake(x, params, handle, all_maps, dtype, need_grad):
    runner = _runner(handle)
    prog = runner.program(x, dtype, all_maps, need_grad)
    td = _TORCH_OUT_DTYPE[dtype]
    outs = [torch.empty_strided((t.B, t.C, t.H, t.W), (t.H * t.W * t.ld, 1, t.W * t.ld, t.ld), dtype=td, device=x.device)
            for t in prog.outs]
    return outs + [torch.empty(1, dtype=torch.int64, device="cpu")]


def _backbone_setup(ctx, inputs, output):
    x, params, handle, all_maps, dtype, need_grad = inputs
    ctx.handle, ctx.n_params, ctx.x_requires_grad = handle, len(params), x.requires_grad
    tok = output[-1]
    # under FakeTensorMode (AOTAutograd, torch.compile, export) the token holds no data: reading it would raise
    # DataDependentOutputException; such a context has no run state and its backward says so
    fake = isinstance(tok, torch._subclasses.fake_tensor.FakeTensor)
    ctx.st = _runner(handle)._pending.pop(int(tok), None) if (need_grad and not fake) else None


def _backbone_backward(ctx, gouts):
    st = ctx.st
    if st is None or st.prog.n_bwd == 0:
        raise RuntimeError("this forward was compiled without a backward list (traced / no-grad call)")
    dx, pgrads = _runner(ctx.handle)._run_backward(st, gouts[:-1], ctx.x_requires_grad)
    ctx.st = None
    return dx, list(pgrads[: ctx.n_params]), None, None, None, None


torch.library.register_autograd("vision_toolbox_amd::backbone", _backbone_backward, setup_context=_backbone_setup)
backbone_op = torch.ops.vision_toolbox_amd.backbone


class BackboneRunner:
    def __init__(self, module: nn.Module):
        self.module = module
        self.store = E.ParamStore(module)
        self.cache: dict = {}
        self.handle = next(_HANDLES)
        self._pending: dict = {}  # token -> run state of a forward whose autograd context has not been set up yet
        _RUNNERS[self.handle] = self

    # -- compile ------------------------------------------------------------------
    def program(self, x: torch.Tensor, dtype: int, all_maps: bool, need_grad: bool) -> Program:
        key = (tuple(x.shape), dtype, all_maps, need_grad, x.requires_grad and need_grad,
               self.store.version, mode_signature(self.module, self.store))
        prog = self.cache.get(key)
        if prog is None:
            B, C_, H, W = x.shape
            b = E.Builder(self.store, dtype, self.module.training, need_grad)
            if dtype == N.VT_BF16:
                # module API: the caller may have changed the f32 masters with any optimiser,
                # so the bf16 mirror is refreshed at the head of every forward
                n = self.store.pflat.numel()
                b.emit(N.OP_COPY2D, [(E.PARAMS, 0), (E.MIRROR, 0)], [N.VT_F32, N.VT_BF16, n, 0], [n, n, 1])
            xr = b.input_images(B, C_, H, W, requires_grad=x.requires_grad and need_grad)
            maps = self.module._vt_emit_maps(b, xr)
            outs = maps if all_maps else maps[-1:]
            seeds = []
            if need_grad:
                seeds = b.seed_output_grads(outs)
                b.build_backward()
            prog = Program(b, outs, seeds)
            self.cache[key] = prog
        return prog

    # -- run ----------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, all_maps: bool, compute_dtype: Optional[torch.dtype] = None):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("expected a 4-D NCHW image tensor")
        if not x.is_cuda:
            # dispatch rule: CPU tensors run the module's own nn children with plain torch
            # ops, exactly what the reference does; nothing here touches libvt_amd or oracle/
            maps = self.module._eager_maps(x)
            return list(maps) if all_maps else list(maps[-1:])
        N.lib()  # raises if the extension is missing: a GPU tensor never falls back to the eager path
        dtype = resolve_dtype(x, compute_dtype)
        tracing = torch._C._get_tracing_state() is not None
        with tracing_paused():
            self.store.ensure(x.device)
            params = self.store.params
            need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if tracing or not need_grad:
            # the parameters are read from the module's flat store at run time: a traced / exported graph holds
            # ONE operator node of x and never has to resolve 200 parameter tensors
            outs = backbone_op(x, [], self.handle, all_maps, dtype, False)
        else:
            outs = backbone_op(x, list(params), self.handle, all_maps, dtype, True)
        return list(outs[:-1])

    def _run_forward(self, prog: Program, x: torch.Tensor):
        dev = x.device
        x32 = x.detach()
        if x32.dtype != torch.float32 or not x32.is_contiguous():
            x32 = x32.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            arena = torch.empty(prog.arena_bytes, dtype=torch.uint8, device=dev)
            st = _RunState()
            st.prog, st.arena, st.x_shape = prog, arena, tuple(x.shape)
            s = self.store
            st.bases = prog.bases(arena.data_ptr(), PARAMS=s.pflat.data_ptr(), STATE=s.sflat.data_ptr(),
                                  MIRROR=s.mirror.data_ptr(), COUNTERS=s.nflat.data_ptr(),
                                  INPUT=x32.data_ptr())
            N.run_ops(prog.fwd_ops, prog.n_fwd, st.bases, current_stream_handle())
        outs = [E.tref_to_tensor(arena, t) for t in prog.outs]
        return st, outs

    def _run_backward(self, st: _RunState, gouts, x_requires_grad: bool):
        prog, arena = st.prog, st.arena
        with torch.cuda.device(arena.device):
            for seed, g in zip(prog.seeds, gouts):
                view = E.tref_to_tensor(arena, seed)
                if g is None:
                    view.zero_()
                else:
                    view.copy_(g)
            N.run_ops(prog.bwd_ops, prog.n_bwd, st.bases, current_stream_handle())
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
        dx = None
        if x_requires_grad and prog.input_grad is not None:
            buf = prog.input_grad
            dx = arena[buf.offset : buf.offset + buf.nbytes].view(torch.float32).view(st.x_shape)
        return dx, pgrads
