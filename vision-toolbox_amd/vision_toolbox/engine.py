"""Static launch-list builder and runner for the backbone hot path.

The reference executes a backbone by walking nn.Sequential in Python and letting
ATen/autograd dispatch conv2d, batch_norm, relu_, cat, add one by one
(vision_toolbox/components.py:26-44, backbones/darknet.py:27-28,51-55,
backbones/vovnet.py:50-63).  Here a backbone is compiled ONCE per
(input shape, dtype, mode) into two flat lists of libvt_amd launches (forward,
backward) over one arena; a step is one call into the native executor
(vt_run_ops) or one hipGraph launch.  Backward is written out explicitly per
unit (no autograd inside), which is what allows concat elision, residual adds
folded into epilogues and gradient accumulation folded into the data-gradient
epilogue.

Only CUDA (HIP) tensors reach this module; CPU tensors are served by the modules' own
torch children (program.py dispatch rule) and never by these launch lists.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch
from torch import nn

from . import _native as N

ALIGN = 256

# base ids (index into the `bases` array handed to vt_run_ops)
ARENA, PARAMS, GRADS, STATE, MIRROR, INPUT, COUNTERS, LABELS, MOMENTUM, HYPER, ZERO_F, ZERO_B = range(12)
NUM_BASES = 12

_TORCH_DTYPE = {N.VT_F32: torch.float32, N.VT_BF16: torch.bfloat16}
_ESIZE = {N.VT_F32: 4, N.VT_BF16: 2}
_EPC = {N.VT_F32: 4, N.VT_BF16: 8}


def _round_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


@dataclass(eq=False)
class Buf:
    base: int
    offset: int  # bytes from the base
    nbytes: int
    name: str = ""


@dataclass(eq=False)
class TRef:
    """NHWC activation, possibly a channel slice [coff, coff+C) of a wider buffer."""

    buf: Buf
    B: int
    H: int
    W: int
    C: int
    ld: int  # pixel stride, elements
    coff: int  # channel offset, elements
    dtype: int
    needs_grad: bool = True

    @property
    def M(self) -> int:
        return self.B * self.H * self.W

    @property
    def esize(self) -> int:
        return _ESIZE[self.dtype]

    def addr(self):
        return (self.buf.base, self.buf.offset + self.coff * self.esize)

    def sl(self, c0: int, c: int) -> "TRef":
        assert 0 <= c0 and c0 + c <= self.C
        return TRef(self.buf, self.B, self.H, self.W, c, self.ld, self.coff + c0, self.dtype, self.needs_grad)

    def same_geom(self, o: "TRef") -> bool:
        return (self.B, self.H, self.W, self.C) == (o.B, o.H, o.W, o.C)


class _PSlice:
    """Elements [start, start + n) of a parameter or buffer of the flat store: one group's rows of a grouped convolution's
    filter, its share of the bias and of the BatchNorm vectors (conv_unit, `groups > 1`)."""

    def __init__(self, base, start: int, n: int):
        self.base, self.start, self.n = base, int(start), int(n)
        self.requires_grad = bool(getattr(base, "requires_grad", False))

    def numel(self) -> int:
        return self.n


class ParamStore:
    """Flat f32 storage behind every parameter / buffer of a module tree.

    Parameters keep their identity, names and logical shapes (so state_dict keys and
    OIHW shapes are exactly the reference's, base.py:23-25), but their storage becomes
    a slice of one flat tensor; 4-D conv weights are held channels_last, i.e.
    physically [Cout][kh][kw][Cin], which is the filter image the kernels read.
    """

    def __init__(self, root: nn.Module, order_key=None):
        self.root = root
        self.order_key = order_key  # optional grouping of the flat layout (stable sort key per parameter)
        self.device = None
        self.pflat = self.sflat = self.nflat = self.mirror = None
        self.pad_multiple = 64  # the flat parameter buffer's length is a multiple of this (sharded exchange: 64 * world)
        self.index: dict[int, tuple[int, int, int]] = {}  # id(tensor owner) -> (base, elem offset, numel)
        self.params: list[nn.Parameter] = []
        self._ptrs: list[tuple[torch.Tensor, int]] = []
        self.version = 0

    # -- layout ---------------------------------------------------------------
    def _collect(self):
        params, fbufs, ibufs, seen = [], [], [], set()
        for p in self.root.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        for mod in self.root.modules():
            for name, b in mod._buffers.items():
                if b is None or id(b) in seen:
                    continue
                seen.add(id(b))
                (fbufs if b.is_floating_point() else ibufs).append((mod, name, b))
        if self.order_key is not None:
            params.sort(key=self.order_key)  # stable: registration order inside each group
        return params, fbufs, ibufs

    def stale(self, device) -> bool:
        if self.device != device or self.pflat is None:
            return True
        for t, ptr in self._ptrs:
            if t.data_ptr() != ptr or t.dtype not in (torch.float32, torch.int64):
                return True
        params, fbufs, ibufs = self._collect()
        return len(params) + len(fbufs) + len(ibufs) != len(self._ptrs)

    def ensure(self, device) -> None:
        if not self.stale(device):
            return
        params, fbufs, ibufs = self._collect()
        for p in params:
            if not p.is_floating_point():
                raise TypeError("non floating point parameter")
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += _round_up(p.numel(), 64)
        self.total = total  # elements that belong to parameters (the rest of pflat is padding)
        pflat = torch.zeros(_round_up(max(total, 64), self.pad_multiple), dtype=torch.float32, device=device)
        soffs, stotal = [], 0
        for _, _, b in fbufs:
            soffs.append(stotal)
            stotal += _round_up(b.numel(), 64)
        sflat = torch.zeros(max(stotal, 64), dtype=torch.float32, device=device)
        nflat = torch.zeros(max(len(ibufs), 1) * 2, dtype=torch.int64, device=device)
        self.index.clear()
        self._ptrs = []
        with torch.no_grad():
            for p, off in zip(params, offs):
                n = p.numel()
                seg = pflat[off : off + n]
                if p.dim() == 4:
                    o, i, kh, kw = p.shape
                    view = seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
                else:
                    view = seg.view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                self.index[id(p)] = (PARAMS, off, n)
                self._ptrs.append((p, p.data_ptr()))
            for (mod, name, b), off in zip(fbufs, soffs):
                n = b.numel()
                view = sflat[off : off + n].view(b.shape)
                view.copy_(b.to(device=device, dtype=torch.float32))
                mod._buffers[name] = view
                self.index[id(view)] = (STATE, off, n)
                self._ptrs.append((view, view.data_ptr()))
            for k, (mod, name, b) in enumerate(ibufs):
                view = nflat[2 * k : 2 * k + 1].view(b.shape)
                view.copy_(b.to(device=device, dtype=torch.int64))
                mod._buffers[name] = view
                self.index[id(view)] = (COUNTERS, 2 * k, 1)
                self._ptrs.append((view, view.data_ptr()))
        self.params, self.offsets = params, offs
        self.pflat, self.sflat, self.nflat = pflat, sflat, nflat
        self.mirror = torch.zeros(pflat.numel(), dtype=torch.bfloat16, device=device)
        self.device = device
        self.version += 1

    def where(self, t) -> tuple[int, int, int]:
        if isinstance(t, _PSlice):
            base, off, _ = self.index[id(t.base)]
            return base, off + t.start, t.n
        return self.index[id(t)]

    def param_signature(self):
        return tuple(p._version for p in self.params)


# ---------------------------------------------------------------------------------------
# gradient bookkeeping (per forward buffer)
# ---------------------------------------------------------------------------------------
@dataclass
class _GradState:
    gbuf: Optional[Buf] = None
    init: list = field(default_factory=list)  # initialised channel intervals [c0, c1) in buffer coords
    pending: list = field(default_factory=list)  # (c0, c1, TRef addend)

    def covered(self, c0: int, c1: int) -> bool:
        pos = c0
        for a, b in sorted(self.init):
            if a > pos:
                break
            pos = max(pos, b)
        return pos >= c1

    def touches(self, c0: int, c1: int) -> bool:
        return any(a < c1 and c0 < b for a, b in self.init)


class Builder:
    """Emits the forward / backward launch lists of one model instance."""

    def __init__(self, store: ParamStore, dtype: int, training: bool, need_grad: bool,
                 grad_base: int = ARENA, master_mirror_fresh: bool = False, deterministic: Optional[bool] = None):
        self.store = store
        self.dtype = dtype
        self.training = training
        self.need_grad = need_grad
        self.grad_base = grad_base
        self.fwd: list[N.Op] = []
        self.bwd: list[N.Op] = []
        self._cur = self.fwd
        self.arena_top = 0
        # scratch that must be zero before the forward / backward list runs lives in two
        # contiguous regions (own base ids), so each list needs ONE memset
        self.zf_top = self.zb_top = 0
        self.nodes: list[Callable[[], None]] = []  # backward emitters, forward order
        self.gstate: dict[int, _GradState] = {}
        self.param_grad_off: dict[int, int] = {}  # id(param) -> byte offset in grad base
        self.pgrad_bytes = 0
        self.tag = 0
        self.n_units = 0
        self.debug_refs: dict[str, TRef] = {}
        # the fused trainer sets this: the repacked data-gradient filters depend only on the
        # weights, so their (tiny, launch-bound) pack kernels leave the backward critical path
        # and run on the side stream while the forward list executes
        self.hoist_dgrad_packs = False
        # Schedule and decomposition choices, each the winner of an alternating A/B inside the train step (NOTEBOOK.md):
        #  * a unit's filter gradient is released to the side stream BEFORE its data gradient (after it for every unit:
        #    22.20 vs 21.27 ms) -- except for the unit that reads the stem's output: its data gradient, its filter gradient
        #    and the stem's one-pass backward are the HBM-bound tail of a step, and released after the data gradient the
        #    filter gradient runs beside the stem's backward instead (21.41 -> 21.27 ms);
        #  * stem unit (3 -> 32, s1): BatchNorm-backward reduction and filter gradient in one pass, dz never formed, and
        #    its pre-activation z (B x 224 x 224 x 32, the largest tensor of a step) never stored: the conv runs twice
        #    (statistics only, then with the normalise + ReLU epilogue) and backward works from y (vt_stem_bwd.hip);
        #  * 3x3 stride-2 data gradients whose dz has at most 128 channels (the HBM-bound ones) run as ONE depth-to-space
        #    launch instead of four parity-class launches that each re-read dz (256: 21.90 vs 21.70 ms).
        self.dgrad_d2s_maxc = 128
        # deterministic mode: the last sums still made with f32 atomics take an order-free form -- the filter gradients
        # become two-stage (partial tiles stored into slabs of ONE scratch that every layer reuses, then an ordered
        # reducer), the bias column sums and the one-pass stem kernel's correlations go through fixed point; with
        # the fixed-point BatchNorm statistics every gradient and parameter update is then bit-identical from run to run
        self.deterministic = (os.environ.get("VT_DETERMINISTIC", "0") != "0") if deterministic is None else bool(deterministic)
        self.wgrad_slab_mb = 48 if self.deterministic else 0  # 0 = atomics
        self._wgrad_slab = None
        # 1x1 ConvNormAct units as four streaming passes that recompute z = W x instead of storing z and dz
        # (vt_pointwise.hip); off in deterministic mode (its filter gradient leaves through float atomics)
        self.pointwise = not self.deterministic
        # ... where the unit's input is at least this many MB: smaller tensors stay in the memory-side cache between the
        # passes of the unfused path, which is then as fast (measured at batch 256, CSPDarknet-53: 22.40 ms with the 51 MB
        # tensors of stage 2 included, 22.11 without; 23.32 with the pointwise path off).  VT_PW_MIN_MB (the tests set 0:
        # their toy tensors would otherwise never reach these kernels)
        self.pointwise_min_mb = float(os.environ.get("VT_PW_MIN_MB", "80"))
        self._hoisted: list[N.Op] = []
        # Filter gradients of same-shape stride-1 3x3 layers (the DarknetBlock.conv2 units of a stage, darknet.py:23-24;
        # an OSA chain, vovnet.py:41-44) are HELD BACK in backward and released together, as consecutive ops that the
        # executor hands to vt_conv_wgrad_group in launches of <= wgrad_group layers: the CU-owning kernel
        # (vt_wgrad6.hip) then pays its prologue and its f32 atomic flush once per launch (isolated, batch 256:
        # 128 -> 128 @28x28 96 us per layer alone, 64 us in a group of 8; the kernel it replaces: 118 us).  Nothing reads a
        # filter gradient before the optimiser (or the bucket all-reduce, which is cut behind the op that completes the
        # bucket), and the arena keeps every x / dz.  VT_WGRAD_GROUP=1: one layer per launch.
        self.wgrad_group = max(1, min(8, int(os.environ.get("VT_WGRAD_GROUP", "8"))))
        # the 1x1 units' filter gradients held back and grouped too?  Off: 8 layers in one launch of the general kernel are
        # cheaper alone (one atomic flush), but released at the end of a stage they no longer run beside their own units'
        # BatchNorm passes: step 20.31 -> 20.43 ms (NOTEBOOK R5.18).  VT_WGRAD_GROUP_1X1=1 turns it on.
        self.wgrad_group_1x1 = os.environ.get("VT_WGRAD_GROUP_1X1", "0") == "1"
        # ... on the side stream (0) or in line on the main stream (1): the CU-owning kernel shares nothing with the
        # kernels beside it (12 waves x 168 registers, 104 KiB of LDS), so the side stream buys it no overlap
        self.wgrad_inline = os.environ.get("VT_WGRAD_INLINE", "0") != "0"
        # Round 6: a 3x3 stride-1 data gradient that is the ONLY contribution to the gradient of a ConvNormAct unit's
        # output also reduces that unit's BatchNorm backward (vt_conv_dgrad_bnred): the separate reduction pass over
        # d(y) and z disappears (DarknetBlock.conv1 <- conv2, darknet.py:23-28: 23 units of CSPDarknet-53).  The last
        # whole-tensor data-gradient op per gradient buffer is remembered here; any other writer forgets it.
        # VT_FUSE_BNRED=0: the separate passes everywhere.
        # (off by default: the fused epilogue waits for z four times per tile -- 91 us against 66 + 19.5 for the two launches
        #  at 128 channels @28x28, step +0.10 ms; NOTEBOOK R6.4.  VT_FUSE_BNRED=1 turns it on.)
        self.fuse_bnred = os.environ.get("VT_FUSE_BNRED", "0") != "0"
        # BatchNorm backward of a unit as ONE launch (vt_bn_act_bwd_fused: the operands stay in registers between the
        # reduction and the apply pass; the library falls back to the three launches where they do not fit).  Not with
        # SyncBatchNorm (the sums are exchanged between the passes).  OFF by default: alone it equals the three launches
        # (32.5 against 30.8 us at 256 channels @14x14: two grid barriers of ~5 us each eat what the second read costs) and
        # in the step it is 0.3 ms SLOWER -- a launch that owns every CU's register file shares nothing with the
        # filter-gradient stream beside it (NOTEBOOK R6.5).  VT_BN_BWD_FUSED=1 turns it on.
        self.bn_bwd_fused = os.environ.get("VT_BN_BWD_FUSED", "0") != "0"
        self._last_dgrad: dict[int, tuple] = {}  # id(gradient Buf) -> (op, c0, c1)
        # The finalize launches folded into the streaming launches that consume their coefficients (vt_bn_finalize_apply,
        # vt_bn_bwd_finalize_apply): every workgroup of the pass finalizes the channels of its own channel group from the
        # (complete) sums -- nothing is handed over inside the launch -- and the single-workgroup finalize launch between a
        # producer of statistics and its consumer disappears: 114 of the 134 of a CSPDarknet-53 step, 132 with the pointwise
        # passes (pw_units); same-box 20.36 -> 19.79 ms (NOTEBOOK R6.10).  Bit-identical.  Not with SyncBatchNorm (the statistics are exchanged in front of the finalize).
        # VT_BN_FIN_APPLY=0 keeps the separate launches.  (The first form -- the first workgroups finalize and publish, all
        # others poll -- measured 2.3 ms SLOWER, R6.6.)
        self.bn_fin_apply = os.environ.get("VT_BN_FIN_APPLY", "1") != "0"
        # (A third form -- the finalize step as the TAIL of the launch that produces the sums: every workgroup takes a ticket,
        # the last one finalizes, nobody waits -- was bit-identical and NO faster than the launch it replaced, and its unused code
        # at the end of the MFMA kernel cost 0.09 ms per step: removed, NOTEBOOK R6.10.)
        self._wg_expect: dict[tuple, int] = {}   # shape key -> units seen in forward and not yet released
        self._wg_pending: dict[tuple, list] = {}  # shape key -> [(x addr, dz addr, dw addr, desc, ldw)]
        # SyncBatchNorm (configs/base.yaml:22): the trainer all-reduces every layer's statistics
        # between the kernel that accumulates them and the finalize kernel; the finalize kernels
        # are then told the GLOBAL sample count (and scale the affine gradients by 1/world)
        self.bn_world = 1
        self.bn_sync = False  # SyncBatchNorm: the sums are exchanged in front of every finalize step, which stays a launch of its own
        # feature-map inputs of programs that do not start from an image (necks): the runner copies
        # the caller's tensors into these buffers before the forward list and reads their gradients
        # (ext_grads, filled by build_backward) after the backward list
        self.ext_inputs: list[TRef] = []
        self.ext_grads: list[Optional[TRef]] = []

    # -- memory -----------------------------------------------------------------
    def alloc(self, nbytes: int, name: str = "") -> Buf:
        off = self.arena_top
        self.arena_top = _round_up(off + max(nbytes, 1), ALIGN)
        return Buf(ARENA, off, nbytes, name)

    def act(self, B, H, W, C, name="", needs_grad=True) -> TRef:
        buf = self.alloc(B * H * W * C * _ESIZE[self.dtype], name)
        t = TRef(buf, B, H, W, C, C, 0, self.dtype, needs_grad)
        if name:
            self.debug_refs[name] = t  # name -> activation, for tools/debug_*.py
        return t

    def debug_grad_ref(self, name: str) -> Optional[TRef]:
        """the gradient buffer mirroring activation `name` (None if never allocated)."""
        t = self.debug_refs.get(name)
        gs = self.gstate.get(id(t.buf)) if t is not None else None
        if gs is None or gs.gbuf is None:
            return None
        return TRef(gs.gbuf, t.B, t.H, t.W, t.C, t.ld, t.coff, t.dtype)

    def f32(self, n: int, name="") -> Buf:
        return self.alloc(n * 4, name)

    def zeroed_f32(self, n: int, name="", bwd: Optional[bool] = None) -> Buf:
        """f32 scratch zeroed once per run of the list being built (stats / reduction sums)."""
        if bwd is None:
            bwd = self._cur is self.bwd
        nbytes = _round_up(n * 4, ALIGN)
        if bwd:
            buf = Buf(ZERO_B, self.zb_top, n * 4, name)
            self.zb_top += nbytes
        else:
            buf = Buf(ZERO_F, self.zf_top, n * 4, name)
            self.zf_top += nbytes
        return buf

    # -- op emission --------------------------------------------------------------
    def emit(self, kind: int, ptrs=(), ints=(), flts=(), desc: Optional[N.ConvDesc] = None, extra_ints=(),
             side: bool = False):
        op = N.Op()
        op.kind = kind | (N.OP_SIDE_STREAM if side else 0)
        op.tag = self.tag
        for k in range(N.VT_OP_MAX_PTR):
            op.ptr[k].base = -1
        for k, p in enumerate(ptrs):
            if p is None:
                continue
            base, off = p
            op.ptr[k].base = base
            op.ptr[k].offset = off
        if desc is not None:
            C.memmove(C.addressof(op.i), C.addressof(desc), C.sizeof(desc))
            k0 = C.sizeof(desc) // 4
            for k, v in enumerate(extra_ints):
                op.i[k0 + k] = int(v)
        else:
            for k, v in enumerate(ints):
                op.i[k] = int(v)
        for k, v in enumerate(flts):
            op.f[k] = float(v)
        self._cur.append(op)
        return op

    @staticmethod
    def bp(buf: Buf, byte_off: int = 0):
        return (buf.base, buf.offset + byte_off)

    # -- parameters ---------------------------------------------------------------
    def pref(self, t, mirror: bool = False):
        base, off, n = self.store.where(t)
        if mirror:
            assert base == PARAMS
            return (MIRROR, off * 2)
        return (base, off * (8 if base == COUNTERS else 4))

    def pgrad(self, p: nn.Parameter):
        """address of the f32 gradient accumulator of parameter p (None if it needs none)."""
        if not p.requires_grad:
            return None
        if isinstance(p, _PSlice):
            a = self.pgrad(p.base)
            return (a[0], a[1] + 4 * p.start)
        if self.grad_base == GRADS:
            _, off, _ = self.store.where(p)
            return (GRADS, off * 4)
        key = id(p)
        if key not in self.param_grad_off:
            buf = self.zeroed_f32(p.numel(), "pgrad", bwd=True)
            self.param_grad_off[key] = buf.offset
        return (ZERO_B, self.param_grad_off[key])

    # -- gradient bookkeeping -------------------------------------------------------
    def _gs(self, t: TRef) -> _GradState:
        return self.gstate.setdefault(id(t.buf), _GradState())

    def _gref(self, t: TRef) -> TRef:
        gs = self._gs(t)
        if gs.gbuf is None:
            gs.gbuf = self.alloc(t.buf.nbytes, "d_" + t.buf.name)
        return TRef(gs.gbuf, t.B, t.H, t.W, t.C, t.ld, t.coff, t.dtype)

    def _add_into(self, dst: TRef, src: TRef, accumulate: bool):
        """dst (=|+=) src, elementwise (vt_bn_act_apply with unit scale)."""
        self._last_dgrad.pop(id(dst.buf), None)
        self.emit(N.OP_BN_ACT_APPLY,
                  [src.addr(), None, None, dst.addr() if accumulate else None, dst.addr()],
                  [src.ld, dst.ld, dst.ld, dst.C, 0, self.dtype], [dst.M])

    def _flush_pending(self, t: TRef):
        gs = self._gs(t)
        c0, c1 = t.coff, t.coff + t.C
        keep = []
        for a, b, add in gs.pending:
            if a < c1 and c0 < b:
                if not (c0 <= a and b <= c1):
                    # a contribution wider than the requested slice (a grouped convolution reads channel slices of a
                    # tensor whose gradient arrives full width): the part inside is added now, the rest stays pending
                    lo, hi = max(a, c0), min(b, c1)
                    if a < lo:
                        keep.append((a, lo, add.sl(0, lo - a)))
                    if hi < b:
                        keep.append((hi, b, add.sl(hi - a, b - hi)))
                    add, a, b = add.sl(lo - a, hi - lo), lo, hi
                sub = t.sl(a - c0, b - a)
                g = self._gref(sub)
                acc = gs.covered(a, b)
                if not acc and gs.touches(a, b):
                    raise NotImplementedError("partially initialised gradient slice")
                self._add_into(g, add, acc)
                if not acc:
                    gs.init.append((a, b))
            else:
                keep.append((a, b, add))
        gs.pending = keep

    def grad_add(self, t: TRef, addend: TRef):
        """register an identity contribution d(t) += addend (materialised lazily)."""
        if not t.needs_grad:
            return
        assert t.same_geom(addend)
        self._gs(t).pending.append((t.coff, t.coff + t.C, addend))

    def grad_target(self, t: TRef):
        """for a kernel that writes d(t) and can fold ONE addend into its epilogue.
        returns (destination, residual or None)."""
        gs = self._gs(t)
        c0, c1 = t.coff, t.coff + t.C
        g = self._gref(t)
        self._last_dgrad.pop(id(g.buf), None)  # (a writer is coming: whatever wrote the buffer last is no longer the only one)
        if gs.covered(c0, c1):
            self._flush_pending(t)
            return g, g
        if gs.touches(c0, c1):
            raise NotImplementedError("partially initialised gradient slice")
        exact = [p for p in gs.pending if p[0] == c0 and p[1] == c1]
        other = [p for p in gs.pending if p[0] < c1 and c0 < p[1] and not (p[0] == c0 and p[1] == c1)]
        if len(exact) == 1 and not other:
            gs.pending.remove(exact[0])
            gs.init.append((c0, c1))
            return g, exact[0][2]
        gs.init.append((c0, c1))
        if exact or other:
            # destination is written first (no residual), the addends are added afterwards
            self._deferred_flush = t
        return g, None

    def grad_written(self, t: TRef):
        """call after emitting the writer returned by grad_target when it had leftovers."""
        d = getattr(self, "_deferred_flush", None)
        if d is not None:
            self._deferred_flush = None
            self._flush_pending(d)

    def grad_read(self, t: TRef) -> Optional[TRef]:
        """the complete gradient of t, or None if nothing contributed."""
        gs = self._gs(t)
        c0, c1 = t.coff, t.coff + t.C
        over = [p for p in gs.pending if p[0] < c1 and c0 < p[1]]
        if not gs.touches(c0, c1):
            if not over:
                return None
            if len(over) == 1 and over[0][0] <= c0 and c1 <= over[0][1]:
                return over[0][2].sl(c0 - over[0][0], c1 - c0)  # read-only alias of the single contribution
        self._flush_pending(t)
        if not gs.covered(c0, c1):
            # zero-fill the gaps (only dense full-width buffers can be memset)
            if t.ld == t.C and not gs.touches(c0, c1):
                g = self._gref(t)
                self._last_dgrad.pop(id(g.buf), None)
                self.emit(N.OP_MEMSET, [g.addr()], [0], [t.M * t.C * t.esize])
                gs.init.append((c0, c1))
            else:
                raise NotImplementedError("gradient slice only partially produced")
        return self._gref(t)

    # -- input / output plumbing -------------------------------------------------------
    MIX_OFF = 32  # byte offset of the MixUp / CutMix parameter block inside the HYPER buffer

    def input_images(self, B, C_, H, W, requires_grad=False, mix: bool = False) -> TRef:
        """NCHW f32 images (base INPUT) -> NHWC dtype with channels padded to 16 bytes; with `mix` the
        conversion applies MixUp / CutMix from the device parameter block (classifier.py:86-87)."""
        epc = _EPC[self.dtype]
        cpad = _round_up(C_, epc)
        x = self.act(B, H, W, cpad, "images", needs_grad=requires_grad)
        self.emit(N.OP_NCHW_TO_NHWC, [(INPUT, 0), x.addr(), (HYPER, self.MIX_OFF) if mix else None],
                  [B, C_, H, W, cpad, self.dtype])
        x.logical_C = C_
        if requires_grad and self.need_grad:
            dx_buf = self.alloc(B * C_ * H * W * 4, "d_images")
            self.input_grad = dx_buf

            def bwd():
                g = self.grad_read(x)
                if g is None:
                    self.emit(N.OP_MEMSET, [self.bp(dx_buf)], [0], [dx_buf.nbytes])
                else:
                    self.emit(N.OP_NHWC_TO_NCHW, [g.addr(), self.bp(dx_buf)], [g.ld, B, C_, H, W, self.dtype])

            self.nodes.append(bwd)
        return x

    # -- the ConvNormAct unit (reference components.py:13-46) --------------------------
    def _taps(self, k: int, dil: int = 1):
        return [(r * dil, t * dil) for r in range(k) for t in range(k)]

    def _conv_desc(self, x: TRef, Cout, Ho, Wo, s, pad, k, ldy, ldw, flags, ldr=0, dil=1) -> N.ConvDesc:
        d = N.ConvDesc()
        d.dtype = self.dtype
        d.B, d.Hi, d.Wi, d.Cin, d.ldx = x.B, x.H, x.W, x.C, x.ld
        d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = Ho, Wo, s, s, -pad, -pad
        d.Cout, d.ldy, d.oH, d.oW = Cout, ldy, Ho, Wo
        d.oHs = d.oWs = 1
        d.oh0 = d.ow0 = 0
        d.ldw, d.ldr, d.flags = ldw, ldr, flags
        taps = self._taps(k, dil)
        d.ntaps = len(taps)
        for i, (r, t) in enumerate(taps):
            d.dh[i], d.dw[i] = r, t
        return d

    def conv_unit(self, x: TRef, conv: nn.Conv2d, norm: Optional[nn.Module], relu: bool,
                  residual: Optional[TRef] = None, out: Optional[TRef] = None, name: str = "",
                  pool_out: Optional[TRef] = None) -> TRef:
        """y = [relu]([bn](conv(x))) [+ residual], written to `out` when given.  `pool_out`: MaxPool2d(3, 2, 1) of y goes
        there as well -- from the unit's own normalise pass where it has one (vt_bn_act_apply_pool), and then the unit's
        BatchNorm backward reads the pooled gradient through the arg-max taps instead of a materialised d(y)."""
        self.tag += 1
        self.n_units += 1
        dt, epc = self.dtype, _EPC[self.dtype]
        k, s = conv.kernel_size[0], conv.stride[0]
        pad = conv.padding[0]
        dil = conv.dilation[0]
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1] or \
                conv.dilation[0] != conv.dilation[1] or conv.padding[0] != conv.padding[1]:
            raise NotImplementedError("hot path covers square convolutions (kernel, stride, dilation, padding)")
        if dil * (k - 1) > 127:
            raise NotImplementedError(f"dilation {dil}: tap offsets are 8-bit")
        if conv.groups != 1:
            self.tag -= 1
            self.n_units -= 1
            return self._grouped_unit(x, conv, norm, relu, residual, out, name, pool_out)
        if k * k > N.VT_MAX_TAPS:
            raise NotImplementedError(f"kernel {k}x{k} exceeds {N.VT_MAX_TAPS} taps")
        Cout, Cin_w = conv.out_channels, conv.in_channels
        logical_cin = getattr(x, "logical_C", x.C)
        if logical_cin != Cin_w:
            raise ValueError(f"conv expects {Cin_w} input channels, got {logical_cin}")
        if Cout % epc:
            raise NotImplementedError(f"out_channels={Cout} must be a multiple of {epc} for dtype {dt}")
        has_bn = isinstance(norm, nn.BatchNorm2d) or getattr(norm, "_vt_bn", False)
        if norm is not None and not has_bn and not isinstance(norm, nn.Identity):
            raise NotImplementedError(f"norm {type(norm).__name__} is outside the hot path")
        if has_bn and (norm.momentum is None or not norm.affine or not norm.track_running_stats):
            raise NotImplementedError("BatchNorm2d variants other than the default are outside the hot path")
        relu = int(relu)  # activation code: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 SiLU, 4 GELU (include/vt_amd.h)
        generic_act = relu >= 2  # only the unfused BatchNorm passes implement these (off the Darknet / VoVNet path)
        Ho = (x.H + 2 * pad - dil * (k - 1) - 1) // s + 1
        Wo = (x.W + 2 * pad - dil * (k - 1) - 1) // s + 1
        if Ho <= 0 or Wo <= 0:
            raise ValueError(f"{name}: a {x.H}x{x.W} map is smaller than the dilated {k}x{k} kernel")
        ntaps = k * k
        B = x.B
        M = B * Ho * Wo

        if has_bn:
            spec = (conv, norm, relu, residual, out, name)
            if self._pw_ok(x, [spec]):
                self.tag -= 1  # (pw_units takes its own tag)
                self.n_units -= 1
                y_pw = self.pw_units(x, [spec])[0]
                if pool_out is not None:
                    self.maxpool3x3s2(y_pw, out=pool_out, name=name + ".max_pool")
                return y_pw

        # ---- filter operand ---------------------------------------------------------
        w = conv.weight
        padded = x.C != Cin_w  # stem: 3 channels padded to one 16-byte chunk
        if padded:
            wpack = self.alloc(Cout * ntaps * x.C * _ESIZE[dt], "wpad")
            self.emit(N.OP_MEMSET, [self.bp(wpack)], [0], [wpack.nbytes])
            # (bf16: from the mirror every other filter is read from -- the same rounding of the same master value, and
            #  under the sharded gradient exchange the mirror is what the all-gather refreshes on every rank, whereas the
            #  f32 master of a slice another rank owns goes stale)
            wsrc, wsrc_dt = (self.pref(w, mirror=True), dt) if dt == N.VT_BF16 else (self.pref(w), N.VT_F32)
            self.emit(N.OP_COPY2D, [wsrc, self.bp(wpack)], [wsrc_dt, dt, Cin_w, 0],
                      [Cin_w, x.C, Cout * ntaps])
            wptr = self.bp(wpack)
        elif dt == N.VT_F32:
            wptr = self.pref(w)
        else:
            wptr = self.pref(w, mirror=True)
        ldw = ntaps * x.C

        if out is not None:
            assert (out.B, out.H, out.W, out.C) == (B, Ho, Wo, Cout), "out geometry mismatch"
        if residual is not None:
            assert (residual.B, residual.H, residual.W, residual.C) == (B, Ho, Wo, Cout)

        track = self.need_grad
        # the unit follows ITS BatchNorm's flag (a frozen bn.eval() inside a training model uses the
        # running statistics and leaves them untouched, like nn.BatchNorm2d)
        unit_training = bool(norm.training) if has_bn else self.training
        fused = has_bn and not unit_training and not track and not generic_act
        y = out if out is not None else self.act(B, Ho, Wo, Cout, name + ".y")
        z = None
        coef = None
        stem_fused = (track and padded and has_bn and not fused and residual is None and not generic_act and
                      not x.needs_grad and w.requires_grad and dt == N.VT_BF16 and Cout == 32 and k == 3 and s == 1 and
                      dil == 1 and pad == 1 and x.C == 8 and x.ld == 8 and x.W <= 888 and B * (x.H + 1) * (x.W + 1) < 0x7fff0000)  # (ring in LDS: halo <= 896 rows)
        stem_y = stem_fused and unit_training and x.W <= 824  # (one more step of halo)
        if has_bn:
            coef = self.f32(4 * Cout, "bncoef")  # scale, shift, mean, invstd
            cp = [self.bp(coef, i * Cout * 4) for i in range(4)]
            g, b_, rm, rv = (self.pref(norm.weight), self.pref(norm.bias), self.pref(norm.running_mean),
                             self.pref(norm.running_var))
            nbt = self.pref(norm.num_batches_tracked) if norm.num_batches_tracked is not None else None
        if fused:
            self.emit(N.OP_BN_EVAL_COEFFS, [g, b_, rm, rv, cp[0], cp[1], None, None], [Cout], [norm.eps])
            flags = N.VT_CONV_AFFINE | (N.VT_CONV_RELU if relu else 0) | (N.VT_CONV_RESIDUAL if residual else 0)
            d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, y.ld, ldw, flags, residual.ld if residual else 0, dil=dil)
            self.emit(N.OP_CONV_IGEMM, [x.addr(), wptr, y.addr(), cp[0], cp[1],
                                        residual.addr() if residual else None, None], desc=d)
        elif stem_y:
            y.stem_out = True  # (the unit that reads it releases its filter gradient late: wgrad_late_stem)
            stats = self.zeroed_f32(N.stat_floats(Cout), "stats")
            d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, y.ld, ldw, N.VT_CONV_STATS | N.VT_CONV_NOSTORE, dil=dil)
            self.emit(N.OP_CONV_IGEMM, [x.addr(), wptr, None, None, None, None, self.bp(stats)], desc=d)
            self.emit(N.OP_BN_FINALIZE,
                      [self.bp(stats), g, b_, rm, rv, nbt, *cp],
                      [Cout], [M * self.bn_world, norm.eps, norm.momentum])
            d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, y.ld, ldw, N.VT_CONV_AFFINE | (N.VT_CONV_RELU if relu else 0), dil=dil)
            self.emit(N.OP_CONV_IGEMM, [x.addr(), wptr, y.addr(), cp[0], cp[1], None, None], desc=d)
        elif has_bn:
            z = self.act(B, Ho, Wo, Cout, name + ".z")
            if unit_training:
                stats = self.zeroed_f32(N.stat_floats(Cout), "stats")
                d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, z.ld, ldw, N.VT_CONV_STATS, dil=dil)
                fin_fwd = (self.bn_fin_apply and not self.bn_sync and not generic_act and
                           not (pool_out is not None and not generic_act))
                self.emit(N.OP_CONV_IGEMM, [x.addr(), wptr, z.addr(), None, None, None, self.bp(stats)], desc=d)
                if not fin_fwd:
                    self.emit(N.OP_BN_FINALIZE,
                              [self.bp(stats), g, b_, rm, rv, nbt, *cp],
                              [Cout], [M * self.bn_world, norm.eps, norm.momentum])
            else:
                fin_fwd = False
                d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, z.ld, ldw, 0, dil=dil)
                self.emit(N.OP_CONV_IGEMM, [x.addr(), wptr, z.addr(), None, None, None, None], desc=d)
                self.emit(N.OP_BN_EVAL_COEFFS, [g, b_, rm, rv, *cp], [Cout], [norm.eps])
            pool_am = None
            if pool_out is not None and not generic_act:
                assert (pool_out.B, pool_out.H, pool_out.W, pool_out.C) == (B, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, Cout)
                pool_am = self.alloc(B * pool_out.H * pool_out.W * Cout, "argmax")
                self.emit(N.OP_BN_ACT_APPLY,
                          [z.addr(), cp[0], cp[1], residual.addr() if residual else None, y.addr(), pool_out.addr(),
                           self.bp(pool_am)],
                          [z.ld, residual.ld if residual else 0, y.ld, Cout, int(relu), dt, pool_out.ld, B, Ho, Wo], [M])
            elif fin_fwd:
                self.emit(N.OP_BN_FIN_APPLY,
                          [self.bp(stats), g, b_, rm, rv, nbt, *cp, z.addr(),
                           residual.addr() if residual else None, y.addr()],
                          [Cout, z.ld, residual.ld if residual else 0, y.ld, int(relu), dt],
                          [M * self.bn_world, norm.eps, norm.momentum, M])
            else:
                self.emit(N.OP_BN_ACT_APPLY,
                          [z.addr(), cp[0], cp[1], residual.addr() if residual else None, y.addr()],
                          [z.ld, residual.ld if residual else 0, y.ld, Cout, int(relu), dt], [M])
        elif relu:
            # conv (+bias) -> activation, no BatchNorm: ConvNormAct(norm="none") (components.py:33-36).  The
            # pre-activation is kept (backward needs act'(z)); the activation is the unit-scale form of the normalise pass.
            z = self.act(B, Ho, Wo, Cout, name + ".z")
            d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, z.ld, ldw, N.VT_CONV_AFFINE if conv.bias is not None else 0,
                                dil=dil)
            self.emit(N.OP_CONV_IGEMM,
                      [x.addr(), wptr, z.addr(), None, self.pref(conv.bias) if conv.bias is not None else None,
                       None, None], desc=d)
            self.emit(N.OP_BN_ACT_APPLY,
                      [z.addr(), None, None, residual.addr() if residual else None, y.addr()],
                      [z.ld, residual.ld if residual else 0, y.ld, Cout, int(relu), dt], [M])
        else:
            # plain conv (+bias): ESE gate conv (vovnet.py:24), classifier head (classifier.py:63)
            flags = (N.VT_CONV_AFFINE if conv.bias is not None else 0) | (N.VT_CONV_RESIDUAL if residual else 0)
            d = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, y.ld, ldw, flags, residual.ld if residual else 0, dil=dil)
            self.emit(N.OP_CONV_IGEMM,
                      [x.addr(), wptr, y.addr(), None, self.pref(conv.bias) if conv.bias is not None else None,
                       residual.addr() if residual else None, None], desc=d)

        pool_fused = pool_out is not None and has_bn and not fused and not stem_y and not generic_act
        if pool_out is not None and not pool_fused:
            self.maxpool3x3s2(y, out=pool_out, name=name + ".max_pool")  # (no normalise pass to fuse it into)

        wg_key = None
        # (round 5: the 1x1 stride-1 units of a stage too -- DarknetBlock.conv1, darknet.py:23 -- through the general
        #  kernel's grouped launch: a launch of one such layer is mostly its atomic flush and its ramp)
        grp3 = k == 3 and s == 1 and dil == 1 and pad == 1 and x.C > 32 and Cout > 32
        grp1 = k == 1 and s == 1 and pad == 0 and self.wgrad_group_1x1
        if (track and has_bn and not fused and not stem_fused and not padded and w.requires_grad and dt == N.VT_BF16 and
                (grp3 or grp1) and self.wgrad_group > 1 and not self.deterministic):
            wg_key = (k, B, x.H, x.W, x.C, x.ld, Cout, ldw)
            self._wg_expect[wg_key] = self._wg_expect.get(wg_key, 0) + 1

        if track:
            tag = self.tag
            training = unit_training

            def bwd():
                self.tag = tag
                pool_grad = None  # (fused pool: the gradient of the pooled map, read through the arg-max taps)
                if pool_fused:
                    dp = self.grad_read(pool_out)
                    if dp is not None:
                        if residual is None and not stem_fused and self.grad_read(y) is None:
                            pool_grad = dp  # the pool is y's only consumer: d(y) is never formed
                        else:  # y feeds something else too (a returned feature map, a shortcut): form d(y) as the pool's backward would
                            gy, res_ = self.grad_target(y)
                            acc = 0
                            if res_ is not None:
                                if res_ is gy or (res_.buf is gy.buf and res_.coff == gy.coff):
                                    acc = 1
                                else:
                                    self._add_into(gy, res_, False)
                                    acc = 1
                            self.emit(N.OP_MAXPOOL_BWD, [dp.addr(), self.bp(pool_am), gy.addr()],
                                      [dp.ld, gy.ld, B, Ho, Wo, Cout, acc, dt])
                            self.grad_written(y)
                dy = self.grad_read(y) if pool_grad is None else None
                if dy is None and pool_grad is None:
                    return
                if residual is not None:
                    self.grad_add(residual, dy)
                if stem_fused:
                    # dz = a*g - b*z + d feeds nothing but the filter gradient (the unit's input is the image): the
                    # pass that reduces (sum g, sum g*xhat) also correlates g, z and 1 with the tap-shifted x, and a
                    # small kernel finishes dW = a*G - b*Z + d*X once (a, b, d) exist (vt_stem_bwd.hip)
                    sums = self.zeroed_f32(N.stat_floats(Cout), "bwdsums")
                    gzx = self.zeroed_f32(N.lib().vt_stem_bn_bwd_scratch_bytes(Cout) // 4, "stem_gzx")
                    zy = y if stem_y else z
                    self.emit(N.OP_STEM_BWD_REDUCE,
                              [x.addr(), dy.addr(), zy.addr(), cp[0], cp[1], cp[2], cp[3], self.bp(sums), self.bp(gzx)],
                              [dt, B, x.H, x.W, Cout, dy.ld, zy.ld, int(relu), int(self.deterministic) | (2 if stem_y else 0)])
                    if stem_y:  # sum g * xhat from the correlations (z is linear in the patch): nothing recovered from y
                        self.emit(N.OP_STEM_BWD_S2, [self.bp(gzx), wptr, cp[2], cp[3], self.bp(sums)],
                                  [Cout, int(self.deterministic)])
                    bcoef = self.f32(3 * Cout, "bwdcoef")
                    self.emit(N.OP_BN_BWD_FINALIZE,
                              [self.bp(sums), cp[0], cp[2], cp[3], self.pgrad(norm.weight), self.pgrad(norm.bias),
                               self.bp(bcoef)], [Cout, int(training)], [M * self.bn_world, 1.0 / self.bn_world])
                    self.emit(N.OP_STEM_BWD_COMBINE, [self.bp(gzx), self.bp(bcoef), self.pgrad(w), wptr if stem_y else None],
                              [Cout, Cin_w, int(self.deterministic)])
                    return
                if has_bn:
                    sums = self.zeroed_f32(N.stat_floats(Cout), "bwdsums")
                    g_ = pool_grad if pool_grad is not None else dy
                    am_ = [self.bp(pool_am)] if pool_grad is not None else []
                    geo = [B, Ho, Wo] if pool_grad is not None else []
                    rec = self._last_dgrad.get(id(dy.buf)) if (self.fuse_bnred and pool_grad is None and dt == N.VT_BF16 and
                                                                not generic_act and self._cur is self.bwd) else None
                    fused_red = rec is not None and rec[1] == dy.coff and rec[2] == dy.coff + dy.C and any(o is rec[0] for o in self.bwd)
                    one_launch = (self.bn_bwd_fused and not fused_red and pool_grad is None and dt == N.VT_BF16 and
                                  not generic_act and not self.bn_sync)
                    bcoef = self.f32(3 * Cout, "bwdcoef")
                    dz = self.act(B, Ho, Wo, Cout, name + ".dz")
                    if one_launch:
                        sync = self.zeroed_f32(4, "bwdsync")
                        self.emit(N.OP_BN_BWD_FUSED,
                                  [dy.addr(), z.addr(), cp[0], cp[1], cp[2], cp[3], self.bp(sums), self.bp(sync),
                                   self.pgrad(norm.weight), self.pgrad(norm.bias), self.bp(bcoef), dz.addr()],
                                  [dy.ld, z.ld, dz.ld, Cout, int(relu), dt, int(training)], [M, M * self.bn_world, 1.0 / self.bn_world])
                    else:
                        if fused_red:
                            # d(y) came out of ONE data-gradient launch and nothing was added to it since: that launch also forms
                            # this unit's backward sums (the op is patched in place: ptr dz w dy | z scale shift mean invstd sums)
                            fop = rec[0]
                            fop.kind = N.OP_CONV_DGRAD_BNRED | (fop.kind & N.OP_SIDE_STREAM)
                            for k_, pa in ((3, z.addr()), (4, cp[0]), (5, cp[1]), (6, cp[2]), (7, cp[3]), (8, self.bp(sums))):
                                fop.ptr[k_].base, fop.ptr[k_].offset = pa
                            k0 = C.sizeof(N.ConvDesc) // 4
                            fop.i[k0], fop.i[k0 + 1] = z.ld, int(relu)
                            self._last_dgrad.pop(id(dy.buf), None)
                        else:
                            self.emit(N.OP_BN_BWD_REDUCE,
                                      [g_.addr(), z.addr(), cp[0], cp[1], cp[2], cp[3], self.bp(sums)] + am_,
                                      [g_.ld, z.ld, Cout, int(relu), dt] + geo, [M])
                        if self.bn_fin_apply and not self.bn_sync and pool_grad is None and not generic_act:
                            self.emit(N.OP_BN_BWD_FIN_APPLY,
                                      [self.bp(sums), cp[0], cp[1], cp[2], cp[3], self.pgrad(norm.weight), self.pgrad(norm.bias),
                                       self.bp(bcoef), g_.addr(), z.addr(), dz.addr()],
                                      [Cout, int(training), g_.ld, z.ld, dz.ld, int(relu), dt],
                                      [M * self.bn_world, 1.0 / self.bn_world, M])
                        else:
                            self.emit(N.OP_BN_BWD_FINALIZE,
                                      [self.bp(sums), cp[0], cp[2], cp[3], self.pgrad(norm.weight), self.pgrad(norm.bias),
                                       self.bp(bcoef)], [Cout, int(training)], [M * self.bn_world, 1.0 / self.bn_world])
                            self.emit(N.OP_BN_BWD_APPLY,
                                      [g_.addr(), z.addr(), cp[0], cp[1], self.bp(bcoef), dz.addr()] + am_,
                                      [g_.ld, z.ld, dz.ld, Cout, int(relu), dt] + geo, [M])
                else:
                    dz = dy
                    if relu:  # dz = dy * act'(z)
                        dz = self.act(B, Ho, Wo, Cout, name + ".dz")
                        self.emit(N.OP_BN_BWD_APPLY, [dy.addr(), z.addr(), None, None, None, dz.addr()],
                                  [dy.ld, z.ld, dz.ld, Cout, int(relu), dt], [M])
                    if conv.bias is not None and conv.bias.requires_grad:
                        if self.deterministic:
                            qb = self.zeroed_f32(4 * Cout, "dbq", bwd=True)
                            self.emit(N.OP_COLSUM, [dz.addr(), self.bp(qb)], [dz.ld, Cout, dt, 1], [M])
                            self.emit(N.OP_FIXED_TO_F32, [self.bp(qb), self.pgrad(conv.bias)], [1], [Cout])
                        else:
                            self.emit(N.OP_COLSUM, [dz.addr(), self.pgrad(conv.bias)], [dz.ld, Cout, dt], [M])
                # filter gradient: needs only x and dz and nothing in backward waits for it, so it goes to the
                # side stream, beside the HBM-bound BatchNorm passes of the units that follow in backward order
                def emit_wgrad():
                    if not w.requires_grad:
                        return
                    dfwd = self._conv_desc(x, Cout, Ho, Wo, s, pad, k, dz.ld, ldw, 0, dil=dil)
                    if wg_key is not None:
                        self._wgrad_hold(wg_key, x.addr(), dz.addr(), self.pgrad(w), dfwd, ldw)
                        return
                    self.emit(N.OP_FORK)
                    slab, slab_mb = None, 0
                    if self.wgrad_slab_mb > 0:  # two-stage (stored slabs + ordered reducer) instead of f32 atomics
                        if self._wgrad_slab is None:
                            self._wgrad_slab = self.alloc(self.wgrad_slab_mb << 20, "wgrad_slabs")
                        slab, slab_mb = self.bp(self._wgrad_slab), self.wgrad_slab_mb
                    if padded:
                        ws = self.zeroed_f32(Cout * ntaps * x.C, "dwpad", bwd=True)
                        self.emit(N.OP_CONV_WGRAD, [x.addr(), dz.addr(), self.bp(ws), slab], desc=dfwd,
                                  extra_ints=[ldw, slab_mb], side=True)
                        self.emit(N.OP_COPY2D, [self.bp(ws), self.pgrad(w)], [N.VT_F32, N.VT_F32, Cin_w, 1],
                                  [x.C, Cin_w, Cout * ntaps], side=True)
                    else:
                        self.emit(N.OP_CONV_WGRAD, [x.addr(), dz.addr(), self.pgrad(w), slab], desc=dfwd,
                                  extra_ints=[ldw, slab_mb], side=True)

                # Released BEFORE the unit's data gradient, except behind the stem (see __init__): there the data gradient
                # runs first and the filter gradient beside the stem's one-pass backward.
                late = x.needs_grad and getattr(x, "stem_out", False)
                if not late:
                    emit_wgrad()
                # data gradient
                if x.needs_grad:
                    self._dgrad(x, dz, wptr if not padded else self.bp(wpack), dt if (padded or dt != N.VT_F32) else N.VT_F32,
                                ldw, Cout, k, s, pad, Ho, Wo, dil)
                if late:
                    emit_wgrad()

            self.nodes.append(bwd)
        return y

    def _grouped_unit(self, x: TRef, conv, norm, relu, residual, out, name, pool_out) -> TRef:
        """nn.Conv2d(groups=G) inside a ConvNormAct (reference components.py:32): G independent units over channel slices
        of x and y -- the kernels take a pixel stride and a channel offset, the filter rows of a group are contiguous in
        the [Cout][kh][kw][Cin / G] image, and BatchNorm is per channel, so a group's statistics, coefficients and
        parameter gradients are slices too.  Slices are addressed in 16-byte chunks: Cin / G and Cout / G must be
        multiples of 8 (bf16) / 4 (f32); depthwise convolutions are outside the Darknet / VoVNet path."""
        import types

        G = conv.groups
        Cin, Cout = conv.in_channels, conv.out_channels
        ci, co = Cin // G, Cout // G
        epc = _EPC[self.dtype]
        if getattr(x, "logical_C", x.C) != Cin or x.C != Cin:
            raise ValueError(f"conv expects {Cin} (unpadded) input channels, got {getattr(x, 'logical_C', x.C)}")
        if ci == 1 and co == 1 and Cin % epc == 0:
            return self._depthwise_unit(x, conv, norm, relu, residual, out, name, pool_out)
        if ci % epc or co % epc:
            raise NotImplementedError(
                f"groups={G} with {ci} -> {co} channels per group: channel slices are addressed in 16-byte chunks "
                f"({epc} elements); narrow groups other than depthwise (groups = in_channels = out_channels) are outside the "
                "hot path")
        k, s, pad, dil = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        Ho = (x.H + 2 * pad - dil * (k - 1) - 1) // s + 1
        Wo = (x.W + 2 * pad - dil * (k - 1) - 1) // s + 1
        y = out if out is not None else self.act(x.B, Ho, Wo, Cout, name + ".y")
        assert (y.B, y.H, y.W, y.C) == (x.B, Ho, Wo, Cout), "out geometry mismatch"
        has_bn = isinstance(norm, nn.BatchNorm2d)
        wn = co * k * k * ci
        for g in range(G):
            cg = types.SimpleNamespace(
                kernel_size=conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation, groups=1,
                in_channels=ci, out_channels=co, weight=_PSlice(conv.weight, g * wn, wn),
                bias=_PSlice(conv.bias, g * co, co) if conv.bias is not None else None, _vt_slice=True)
            ng = norm
            if has_bn:
                ng = types.SimpleNamespace(
                    weight=_PSlice(norm.weight, g * co, co), bias=_PSlice(norm.bias, g * co, co),
                    running_mean=_PSlice(norm.running_mean, g * co, co), running_var=_PSlice(norm.running_var, g * co, co),
                    # (the batch counter belongs to the module, not to a group: the first group's finalize advances it)
                    num_batches_tracked=norm.num_batches_tracked if g == 0 else None,
                    eps=norm.eps, momentum=norm.momentum, affine=norm.affine, track_running_stats=norm.track_running_stats,
                    training=norm.training, _vt_bn=True)
            self.conv_unit(x.sl(g * ci, ci), cg, ng, relu,
                           residual=residual.sl(g * co, co) if residual is not None else None,
                           out=y.sl(g * co, co), name=f"{name}.g{g}")
        if pool_out is not None:
            self.maxpool3x3s2(y, out=pool_out, name=name + ".max_pool")
        return y

    def _depthwise_unit(self, x: TRef, conv, norm, relu, residual, out, name, pool_out) -> TRef:
        """nn.Conv2d(C, C, k, groups=C) inside a ConvNormAct (reference components.py:26-44 with `groups = in_channels`):
        vt_dwconv_fwd (+ batch statistics) -> the ordinary BatchNorm finalize / normalise passes; backward: the ordinary
        BatchNorm backward -> vt_dwconv_wgrad (side stream) and vt_dwconv_dgrad.  Streaming kernels (round 6): off the
        Darknet / VoVNet path, present so that every `groups` the constructor accepts runs on the GPU."""
        Cc = conv.in_channels
        k, s, pad, dil = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        B = x.B
        Ho = (x.H + 2 * pad - dil * (k - 1) - 1) // s + 1
        Wo = (x.W + 2 * pad - dil * (k - 1) - 1) // s + 1
        M, dt = B * Ho * Wo, self.dtype
        if self.deterministic:
            raise NotImplementedError("depthwise filter gradients use f32 atomics: not available in deterministic mode")
        y = out if out is not None else self.act(B, Ho, Wo, Cc, name + ".y")
        assert (y.B, y.H, y.W, y.C) == (B, Ho, Wo, Cc), "out geometry mismatch"
        has_bn = isinstance(norm, nn.BatchNorm2d)
        # (the unit follows ITS BatchNorm's flag, as conv_unit does)
        training = bool(norm.training) if has_bn else self.training
        track = self.need_grad
        self.tag += 1
        self.n_units += 1
        w = conv.weight
        wptr = self.pref(w)  # the f32 master [C][k*k] (the kernels round it for bf16 launches)
        geo = [B, x.H, x.W, Cc, k, s, pad, dil, dt]
        z = self.act(B, Ho, Wo, Cc, name + ".z") if (has_bn or relu) else y
        cp = None
        if has_bn:
            coef = self.f32(4 * Cc, "bncoef")
            cp = [self.bp(coef, i * Cc * 4) for i in range(4)]
            g, b_, rm, rv = (self.pref(norm.weight), self.pref(norm.bias), self.pref(norm.running_mean),
                             self.pref(norm.running_var))
            nbt = self.pref(norm.num_batches_tracked) if norm.num_batches_tracked is not None else None
            if training:
                stats = self.zeroed_f32(N.stat_floats(Cc), "stats")
                self.emit(N.OP_DWCONV_FWD, [x.addr(), wptr, z.addr(), self.bp(stats)], [x.ld, z.ld, 0] + geo)
                self.emit(N.OP_BN_FINALIZE, [self.bp(stats), g, b_, rm, rv, nbt, *cp], [Cc],
                          [M * self.bn_world, norm.eps, norm.momentum])
            else:
                self.emit(N.OP_DWCONV_FWD, [x.addr(), wptr, z.addr(), None], [x.ld, z.ld, 0] + geo)
                self.emit(N.OP_BN_EVAL_COEFFS, [g, b_, rm, rv, *cp], [Cc], [norm.eps])
            self.emit(N.OP_BN_ACT_APPLY, [z.addr(), cp[0], cp[1], residual.addr() if residual else None, y.addr()],
                      [z.ld, residual.ld if residual else 0, y.ld, Cc, int(relu), dt], [M])
        else:
            self.emit(N.OP_DWCONV_FWD, [x.addr(), wptr, z.addr(), None], [x.ld, z.ld, 0] + geo)
            bias = self.pref(conv.bias) if conv.bias is not None else None
            if z is not y or bias is not None or residual is not None:
                # (biased conv -> activation: the unit-scale form of the normalise pass, shift = the bias)
                self.emit(N.OP_BN_ACT_APPLY, [z.addr(), None, bias, residual.addr() if residual else None, y.addr()],
                          [z.ld, residual.ld if residual else 0, y.ld, Cc, int(relu), dt], [M])
        if pool_out is not None:
            self.maxpool3x3s2(y, out=pool_out, name=name + ".max_pool")
        if track:
            tag = self.tag

            def bwd():
                self.tag = tag
                dy = self.grad_read(y)
                if dy is None:
                    return
                if residual is not None:
                    self.grad_add(residual, dy)
                if has_bn:
                    sums = self.zeroed_f32(N.stat_floats(Cc), "bwdsums")
                    self.emit(N.OP_BN_BWD_REDUCE, [dy.addr(), z.addr(), cp[0], cp[1], cp[2], cp[3], self.bp(sums)],
                              [dy.ld, z.ld, Cc, int(relu), dt], [M])
                    bcoef = self.f32(3 * Cc, "bwdcoef")
                    self.emit(N.OP_BN_BWD_FINALIZE,
                              [self.bp(sums), cp[0], cp[2], cp[3], self.pgrad(norm.weight), self.pgrad(norm.bias), self.bp(bcoef)],
                              [Cc, int(training)], [M * self.bn_world, 1.0 / self.bn_world])
                    dz = self.act(B, Ho, Wo, Cc, name + ".dz")
                    self.emit(N.OP_BN_BWD_APPLY, [dy.addr(), z.addr(), cp[0], cp[1], self.bp(bcoef), dz.addr()],
                              [dy.ld, z.ld, dz.ld, Cc, int(relu), dt], [M])
                else:
                    dz = dy
                    if relu:  # dz = dy * act'(z + bias): the coefficient-free form reads the pre-activation it is given
                        zb = z
                        if conv.bias is not None:  # act' is taken at z + bias: form it once
                            zb = self.act(B, Ho, Wo, Cc, name + ".zb")
                            self.emit(N.OP_BN_ACT_APPLY, [z.addr(), None, self.pref(conv.bias), None, zb.addr()],
                                      [z.ld, 0, zb.ld, Cc, 0, dt], [M])
                        dz = self.act(B, Ho, Wo, Cc, name + ".dz")
                        self.emit(N.OP_BN_BWD_APPLY, [dy.addr(), zb.addr(), None, None, None, dz.addr()],
                                  [dy.ld, zb.ld, dz.ld, Cc, int(relu), dt], [M])
                    if conv.bias is not None and conv.bias.requires_grad:
                        self.emit(N.OP_COLSUM, [dz.addr(), self.pgrad(conv.bias)], [dz.ld, Cc, dt], [M])
                if w.requires_grad:
                    self.emit(N.OP_FORK)
                    self.emit(N.OP_DWCONV_WGRAD, [x.addr(), dz.addr(), self.pgrad(w)], [x.ld, dz.ld, 0] + geo, side=True)
                if x.needs_grad:
                    gx, res = self.grad_target(x)
                    self.emit(N.OP_DWCONV_DGRAD, [dz.addr(), wptr, gx.addr(), res.addr() if res is not None else None],
                              [dz.ld, gx.ld, res.ld if res is not None else 0] + geo)
                    self.grad_written(x)

            self.nodes.append(bwd)
        return y

    # -- pointwise (1x1) units without stored pre-activations (vt_pointwise.hip) ------------------------------------
    def _pw_ok(self, x: TRef, specs) -> int:
        """0: the pointwise kernels do not apply to these units (reading the same x); 2: they do, filter gradient
        included; 1: they do, with dz handed to the filter-gradient kernel (vt_pw_supported)."""
        if not self.pointwise or self.dtype != N.VT_BF16 or not (1 <= len(specs) <= 2):
            return 0
        flags = set()
        for conv, norm, relu, residual, out, _ in specs:
            if getattr(conv, "_vt_slice", False):  # (one group of a grouped convolution: parameter slices)
                return 0
            if (conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.dilation != (1, 1)
                    or conv.groups != 1 or conv.bias is not None or not isinstance(norm, nn.BatchNorm2d)
                    or norm.momentum is None or not norm.affine or not norm.track_running_stats):
                return 0
            if conv.in_channels != x.C or getattr(x, "logical_C", x.C) != x.C:
                return 0
            if int(relu) >= 2:
                return 0  # (LeakyReLU / SiLU / GELU: the unfused BatchNorm passes only)
            flags.add((bool(norm.training), bool(relu)))
        if len(flags) != 1 or len({sp[3] is None for sp in specs}) != 1:  # (a residual for every group or for none)
            return 0
        unit_training, _ = next(iter(flags))
        # (inference: the apply pass alone, with the running-statistics coefficients -- it streams x once at ~5.5 TB/s
        #  where the conv launch with the affine + ReLU epilogue stages it through LDS at ~2.7: 160 -> 160 @80x80 x 64
        #  images, 46 vs 91 us)
        if self.need_grad and not x.needs_grad:
            return 0  # (the backward kernel always forms dx)
        if x.M * x.C * 2 < self.pointwise_min_mb * 1e6:
            return 0
        cs = [sp[0].out_channels for sp in specs] + [0]
        mode = int(N.lib().vt_pw_supported(self.dtype, x.C, cs[0], cs[1]))
        if mode == 0 and len(specs) == 1 and not unit_training and not self.need_grad:
            # inference runs the apply pass alone: also on the 80-channel shapes of YOLOv5x's first stage
            mode = 1 if N.lib().vt_pw_apply_supported(self.dtype, x.C, cs[0]) else 0
        return mode

    def pw_units(self, x: TRef, specs) -> "list[TRef]":
        """one or two 1x1 ConvNormAct units reading the same tensor x (components.py:26-44; two: CSPDarknetStage's
        conv1 | conv2, darknet.py:46-47,53) as ONE launch per pass: statistics, normalise, backward reduction, backward
        apply with the data gradient and (small shapes) the filter gradient.  z and dz are never stored."""
        mode = self._pw_ok(x, specs)
        assert mode in (1, 2)
        self.tag += 1
        self.n_units += len(specs)
        dt = self.dtype
        G = len(specs)
        Cs = [sp[0].out_channels for sp in specs]
        Ntot, K, M = sum(Cs), x.C, x.M
        offs = [0, Cs[0]][:G]
        relu = bool(specs[0][2])
        unit_training = bool(specs[0][1].training)
        convs, norms = [sp[0] for sp in specs], [sp[1] for sp in specs]
        ys = []
        for (conv, norm, _, residual, out, name), c in zip(specs, Cs):
            if out is not None:
                assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, c), "out geometry mismatch"
            if residual is not None:
                assert (residual.B, residual.H, residual.W, residual.C) == (x.B, x.H, x.W, c)
            ys.append(out if out is not None else self.act(x.B, x.H, x.W, c, name + ".y"))
        coef = self.f32(4 * Ntot, "bncoef4")  # scale | shift | mean | invstd over all groups' channels
        cps = [[self.bp(coef, (i * Ntot + o) * 4) for i in range(4)] for o in offs]
        wps = [self.pref(c.weight, mirror=True) for c in convs]
        pad2 = lambda v, fill=None: list(v) + [fill] * (2 - len(v))
        head_i = [K, G, int(relu)] + pad2(Cs, 0) + [x.ld] + pad2([K] * G, 0)
        # the finalize step of every group inside the apply passes (vt_pw_fwd_apply_finalize / vt_pw_bwd_apply_finalize)
        pw_fin = self.bn_fin_apply and not self.bn_sync and Ntot <= 128
        fin_p = []
        if unit_training:
            stats = [self.zeroed_f32(N.stat_floats(c), "stats") for c in Cs]
            self.emit(N.OP_PW_STATS, [x.addr(), *pad2(wps), *pad2([self.bp(s_) for s_ in stats])], head_i, [M])
            for g, norm in enumerate(norms):
                if pw_fin:
                    fin_p += [self.bp(stats[g]), self.pref(norm.weight), self.pref(norm.bias), self.pref(norm.running_mean),
                              self.pref(norm.running_var), self.pref(norm.num_batches_tracked)]
                    continue
                self.emit(N.OP_BN_FINALIZE,
                          [self.bp(stats[g]), self.pref(norm.weight), self.pref(norm.bias), self.pref(norm.running_mean),
                           self.pref(norm.running_var), self.pref(norm.num_batches_tracked), *cps[g]],
                          [Cs[g]], [M * self.bn_world, norm.eps, norm.momentum])
        else:
            for g, norm in enumerate(norms):
                self.emit(N.OP_BN_EVAL_COEFFS, [self.pref(norm.weight), self.pref(norm.bias), self.pref(norm.running_mean),
                                                self.pref(norm.running_var), *cps[g]], [Cs[g]], [norm.eps])
        ress = [sp[3] for sp in specs]
        if fin_p:
            epsmom = [v for norm in norms for v in (norm.eps, norm.momentum)]
            self.emit(N.OP_PW_APPLY_FIN,
                      [x.addr(), *pad2(wps), self.bp(coef), *pad2([y.addr() for y in ys]),
                       *pad2([r.addr() if r is not None else None for r in ress]), *fin_p],
                      head_i + pad2([y.ld for y in ys], 0) + pad2([r.ld if r is not None else 0 for r in ress], 0),
                      [M, M * self.bn_world] + epsmom)
        else:
            self.emit(N.OP_PW_APPLY,
                      [x.addr(), *pad2(wps), self.bp(coef), *pad2([y.addr() for y in ys]),
                       *pad2([r.addr() if r is not None else None for r in ress])],
                      head_i + pad2([y.ld for y in ys], 0) + pad2([r.ld if r is not None else 0 for r in ress], 0), [M])
        if self.need_grad:
            tag = self.tag

            def bwd():
                self.tag = tag
                dys = [self.grad_read(y) for y in ys]
                if all(d is None for d in dys):
                    return
                for g in range(G):
                    if dys[g] is None:  # a branch nothing back-propagates into: a zero gradient
                        z_ = self.act(x.B, x.H, x.W, Cs[g], specs[g][5] + ".dy0")
                        self.emit(N.OP_MEMSET, [z_.addr()], [0], [z_.buf.nbytes])
                        dys[g] = z_
                    if ress[g] is not None:
                        self.grad_add(ress[g], dys[g])
                sums = [self.zeroed_f32(N.stat_floats(c), "bwdsums") for c in Cs]
                dy_p, dy_ld = pad2([d.addr() for d in dys]), pad2([d.ld for d in dys], 0)
                self.emit(N.OP_PW_REDUCE, [x.addr(), *pad2(wps), self.bp(coef), *dy_p, *pad2([self.bp(s_) for s_ in sums])],
                          head_i + dy_ld, [M])
                bcoefs = [self.f32(3 * c, "bwdcoef") for c in Cs]
                bfin_p = []
                for g, norm in enumerate(norms):
                    if pw_fin:
                        bfin_p += [self.bp(sums[g]), self.pgrad(norm.weight), self.pgrad(norm.bias)]
                        continue
                    self.emit(N.OP_BN_BWD_FINALIZE,
                              [self.bp(sums[g]), cps[g][0], cps[g][2], cps[g][3], self.pgrad(norm.weight), self.pgrad(norm.bias),
                               self.bp(bcoefs[g])], [Cs[g], int(unit_training)], [M * self.bn_world, 1.0 / self.bn_world])
                gx, res = self.grad_target(x)
                want_dw = [c.weight.requires_grad for c in convs]
                dws = [self.pgrad(c.weight) if (mode == 2 and w_) else None for c, w_ in zip(convs, want_dw)]
                dzs = [self.act(x.B, x.H, x.W, c, sp[5] + ".dz") if (mode == 1 and w_) else None
                       for c, sp, w_ in zip(Cs, specs, want_dw)]
                self.emit(N.OP_PW_BWD_FIN if bfin_p else N.OP_PW_BWD,
                          [x.addr(), *pad2(wps), self.bp(coef), *dy_p, *pad2([self.bp(b_) for b_ in bcoefs]), gx.addr(),
                           res.addr() if res is not None else None, *pad2(dws),
                           *pad2([d.addr() if d is not None else None for d in dzs]), *bfin_p],
                          head_i + dy_ld + [gx.ld, res.ld if res is not None else 0] + pad2([K] * G, 0) +
                          pad2([d.ld if d is not None else 0 for d in dzs], 0) + [int(unit_training)],
                          [M, M * self.bn_world, 1.0 / self.bn_world])
                self.grad_written(x)
                if mode == 1 and any(d is not None for d in dzs):
                    # the filter gradient does not fit the kernel's accumulators: dz was written, the usual kernel takes it
                    self.emit(N.OP_FORK)
                    for g, conv in enumerate(convs):
                        if dzs[g] is None:
                            continue
                        dfwd = self._conv_desc(x, Cs[g], x.H, x.W, 1, 0, 1, dzs[g].ld, K, 0)
                        self.emit(N.OP_CONV_WGRAD, [x.addr(), dzs[g].addr(), self.pgrad(conv.weight), None], desc=dfwd,
                                  extra_ints=[K, 0], side=True)

            self.nodes.append(bwd)
        return ys

    def conv_unit_pair(self, x: TRef, a, b):
        """two ConvNormAct units that read the same tensor (CSPDarknetStage.conv1 / conv2): one pointwise launch per
        pass when the kernels cover the joint shape, else two independent units.  a, b = (ConvNormAct, out, name)."""
        def spec(t):
            m, out, name = t
            norm = m.norm if isinstance(m.norm, nn.BatchNorm2d) else None
            return (m.conv, norm, m._vt_relu(), None, out, name)

        sa, sb = spec(a), spec(b)
        if sa[1] is not None and sb[1] is not None and self._pw_ok(x, [sa, sb]):
            return self.pw_units(x, [sa, sb])
        return [a[0]._vt_emit(self, x, out=a[1], name=a[2]), b[0]._vt_emit(self, x, out=b[1], name=b[2])]

    def _wgrad_hold(self, key, xa, dza, dwa, desc, ldw):
        """hold a filter gradient back until its shape group is complete (or wgrad_group of them wait), then release the
        group as consecutive side-stream ops behind ONE fork (the executor gathers them: vt_conv_wgrad_group)."""
        pend = self._wg_pending.setdefault(key, [])
        pend.append((xa, dza, dwa, desc, ldw))
        self._wg_expect[key] -= 1
        if len(pend) >= self.wgrad_group or self._wg_expect[key] <= 0:
            self._wgrad_release(key)

    def _wgrad_release(self, key=None):
        for k_ in ([key] if key is not None else list(self._wg_pending)):
            pend = self._wg_pending.pop(k_, [])
            if not pend:
                continue
            side = not self.wgrad_inline
            if side:
                self.emit(N.OP_FORK)
            for xa, dza, dwa, desc, ldw in pend:
                self.emit(N.OP_CONV_WGRAD, [xa, dza, dwa, None], desc=desc, extra_ints=[ldw, 0], side=side)

    def _dgrad(self, x: TRef, dz: TRef, wptr, w_dtype, ldw, Cout, k, s, pad, Ho, Wo, dil=1):
        dt = self.dtype
        # the s*s parity classes tile d(x) disjointly, so they share one destination and
        # one folded addend: every pixel is produced exactly once
        gx, res = self.grad_target(x)
        if (s == 2 and k == 3 and dil == 1 and pad == 1 and x.H % 2 == 0 and x.W % 2 == 0 and Cout <= self.dgrad_d2s_maxc and
                (4 * x.C) % (4 * _EPC[dt]) == 0 and dt == N.VT_BF16):
            # Every parity class (ph, pw) of d(x) reads dz at offsets {0, 1}^2 of its own grid position, so the four
            # are the column blocks of ONE 2x2-tap convolution over dz with 4*C output columns (zero taps where a
            # class has fewer), written depth-to-space: dz is read once instead of four times.  16/9 of the MFMA work,
            # which these layers (HBM-bound: few channels, large maps) do not notice.
            taps = [(1, 1), (1, 0), (0, 1), (0, 0)]  # the order a parity class lists its own taps in (below)
            wd = self.alloc(4 * x.C * 4 * Cout * _ESIZE[dt], "wd_d2s")
            for ph in range(2):
                for pw in range(2):
                    r0, t0 = (ph + pad) % s, (pw + pad) % s
                    rows, cols = list(range(r0, k, s)), list(range(t0, k, s))
                    eh, ew = (ph + pad - r0) // s, (pw + pad - t0) // s
                    sel = []
                    for (a, b) in taps:  # class tap (u, v) sits at offset (eh - u, ew - v)
                        u, v = eh - a, ew - b
                        sel.append(rows[u] * k + cols[v] if 0 <= u < len(rows) and 0 <= v < len(cols) else -1)
                    ints = [w_dtype, ldw, dt, 4, Cout, k * k, x.C, 0] + sel
                    dst = self.bp(wd, (2 * ph + pw) * x.C * 4 * Cout * _ESIZE[dt])
                    if self.hoist_dgrad_packs:
                        keep, self._cur = self._cur, self._hoisted
                        self.emit(N.OP_PACK_DGRAD, [wptr, dst], ints, side=True)
                        self._cur = keep
                    else:
                        self.emit(N.OP_PACK_DGRAD, [wptr, dst], ints)
            d = N.ConvDesc()
            d.dtype = dt
            d.B, d.Hi, d.Wi, d.Cin, d.ldx = dz.B, Ho, Wo, Cout, dz.ld
            d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = x.H // 2, x.W // 2, 1, 1, 0, 0
            d.Cout, d.ldy, d.oH, d.oW = 4 * x.C, gx.ld, x.H, x.W
            d.oHs, d.oWs, d.oh0, d.ow0 = 2, 2, 0, 0
            d.ldw, d.ldr = 4 * Cout, (res.ld if res is not None else 0)
            d.flags = N.VT_CONV_D2S | (N.VT_CONV_RESIDUAL if res is not None else 0)
            d.ntaps = 4
            for i, (a, b) in enumerate(taps):
                d.dh[i], d.dw[i] = a, b
            self.emit(N.OP_CONV_IGEMM, [dz.addr(), self.bp(wd), gx.addr(), None, None,
                                        res.addr() if res is not None else None, None], desc=d)
            self.grad_written(x)
            return
        # A dilated strided convolution may never read some parity classes of its input (s = 2, dilation 2: the rows
        # 2i - 1 + 2r are all odd): those classes of d(x) are zero, i.e. the folded addend alone.  The whole of d(x) is
        # initialised first then (addend or zeros) and the classes that exist accumulate into it.
        if any(all((ph + pad - r * dil) % s for r in range(k)) for ph in range(s)):
            if res is None:
                zeros = self.act(x.B, x.H, x.W, x.C, "d0")
                self.emit(N.OP_MEMSET, [zeros.addr()], [0], [zeros.M * zeros.C * zeros.esize])
                self._add_into(gx, zeros, False)
            elif not (res.buf is gx.buf and res.coff == gx.coff):
                self._add_into(gx, res, False)
            res = gx
        for ph in range(s):
            for pw in range(s):
                # d(x)[ph + s*c] = sum over the filter rows r with (ph + pad - r*dil) divisible by s of
                # dz[c + (ph + pad - r*dil) / s] * w[r]  (columns alike): a stride-1 convolution over dz per parity class
                rows = [r for r in range(k) if (ph + pad - r * dil) % s == 0]
                cols = [t for t in range(k) if (pw + pad - t * dil) % s == 0]
                Hc = (x.H - ph + s - 1) // s
                Wc = (x.W - pw + s - 1) // s
                if Hc <= 0 or Wc <= 0:
                    continue
                if not rows or not cols:
                    continue  # (initialised above)
                sel = [r * k + t for r in rows for t in cols]
                offs = [((ph + pad - r * dil) // s, (pw + pad - t * dil) // s) for r in rows for t in cols]
                nsel = len(sel)
                wd = self.alloc(x.C * nsel * Cout * _ESIZE[dt], "wd")
                ints = [w_dtype, ldw, dt, nsel, Cout, k * k, x.C, 0] + sel
                if self.hoist_dgrad_packs:
                    keep, self._cur = self._cur, self._hoisted
                    self.emit(N.OP_PACK_DGRAD, [wptr, self.bp(wd)], ints, side=True)
                    self._cur = keep
                else:
                    self.emit(N.OP_PACK_DGRAD, [wptr, self.bp(wd)], ints)
                d = N.ConvDesc()
                d.dtype = dt
                d.B, d.Hi, d.Wi, d.Cin, d.ldx = dz.B, Ho, Wo, Cout, dz.ld
                d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = Hc, Wc, 1, 1, 0, 0
                d.Cout, d.ldy, d.oH, d.oW = x.C, gx.ld, x.H, x.W
                d.oHs, d.oWs, d.oh0, d.ow0 = s, s, ph, pw
                d.ldw, d.ldr = nsel * Cout, (res.ld if res is not None else 0)
                d.flags = N.VT_CONV_RESIDUAL if res is not None else 0
                d.ntaps = nsel
                for i, (a, b) in enumerate(offs):
                    d.dh[i], d.dw[i] = a, b
                op = self.emit(N.OP_CONV_IGEMM, [dz.addr(), self.bp(wd), gx.addr(), None, None,
                                                 res.addr() if res is not None else None, None], desc=d)
                if s == 1 and res is None and Hc == x.H and Wc == x.W and self._cur is self.bwd:
                    self._last_dgrad[id(gx.buf)] = (op, gx.coff, gx.coff + gx.C)  # this launch alone forms d(x)
        self.grad_written(x)

    def copy(self, x: TRef, dst: TRef) -> TRef:
        """dst = x (places a tensor into a channel slice of a wider buffer)."""
        assert x.same_geom(dst)
        self.tag += 1
        self._add_into(dst, x, False)
        if self.need_grad and x.needs_grad:

            def bwd():
                g = self.grad_read(dst)
                if g is not None:
                    self.grad_add(x, g)

            self.nodes.append(bwd)
        return dst

    # -- pooling -------------------------------------------------------------------------
    def input_map(self, B, C_, H, W, name="in", requires_grad=True) -> TRef:
        """an NHWC feature map supplied by the caller (necks.py:83: the backbone's outputs)."""
        t = self.act(B, H, W, C_, name, needs_grad=requires_grad)
        idx = len(self.ext_inputs)
        self.ext_inputs.append(t)
        self.ext_grads.append(None)
        if requires_grad and self.need_grad:

            def bwd():  # first node appended => last to run: every contribution has been registered
                self.ext_grads[idx] = self.grad_read(t)

            self.nodes.append(bwd)
        return t

    def resample_add(self, src: TRef, other: Optional[TRef], mode: int, name="resample", out: Optional[TRef] = None) -> TRef:
        """nearest x2 (mode 0) / x0.5 (mode 1) or bilinear x2 (mode 2) / x0.5 (mode 3) resampling of `src` plus `other`
        (necks.py:66-81); `out`: write into this tensor (a channel slice of a concat buffer: fuse_fn="concat") instead of
        a fresh one."""
        self.tag += 1
        if mode in (0, 2):
            Hd, Wd = src.H * 2, src.W * 2
        else:
            if src.H % 2 or src.W % 2:
                raise NotImplementedError("x0.5 resampling of an odd-sized map")
            Hd, Wd = src.H // 2, src.W // 2
        if other is not None:
            assert (other.B, other.H, other.W, other.C) == (src.B, Hd, Wd, src.C), "fuse operands differ in shape"
        y = out if out is not None else self.act(src.B, Hd, Wd, src.C, name)
        assert (y.B, y.H, y.W, y.C) == (src.B, Hd, Wd, src.C)
        self.emit(N.OP_RESAMPLE_FWD, [src.addr(), other.addr() if other is not None else None, y.addr()],
                  [src.ld, other.ld if other is not None else 0, y.ld, src.B, Hd, Wd, src.C, mode, self.dtype])
        if self.need_grad and (src.needs_grad or (other is not None and other.needs_grad)):
            tag = self.tag

            def bwd():
                self.tag = tag
                dy = self.grad_read(y)
                if dy is None:
                    return
                if other is not None:
                    self.grad_add(other, dy)
                if not src.needs_grad:
                    return
                gx, res = self.grad_target(src)
                acc = 0
                if res is not None:
                    if res is gx or (res.buf is gx.buf and res.coff == gx.coff):
                        acc = 1
                    else:  # a foreign addend: materialise it first, then accumulate
                        self._add_into(gx, res, False)
                        acc = 1
                self.emit(N.OP_RESAMPLE_BWD, [dy.addr(), gx.addr()],
                          [dy.ld, gx.ld, src.B, Hd, Wd, src.C, mode, acc, self.dtype])
                self.grad_written(src)

            self.nodes.append(bwd)
        return y

    def maxpool3x3s2(self, x: TRef, out: Optional[TRef] = None, name="maxpool") -> TRef:
        self.tag += 1
        Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
        y = out if out is not None else self.act(x.B, Ho, Wo, x.C, name)
        assert (y.B, y.H, y.W, y.C) == (x.B, Ho, Wo, x.C)
        am = self.alloc(x.B * Ho * Wo * x.C, "argmax")
        self.emit(N.OP_MAXPOOL_FWD, [x.addr(), y.addr(), self.bp(am)],
                  [x.ld, y.ld, x.B, x.H, x.W, x.C, self.dtype])
        if self.need_grad and x.needs_grad:
            tag = self.tag

            def bwd():
                self.tag = tag
                dy = self.grad_read(y)
                if dy is None:
                    return
                gx, res = self.grad_target(x)
                acc = 0
                if res is not None:
                    if res is gx or (res.buf is gx.buf and res.coff == gx.coff):
                        acc = 1
                    else:  # a foreign addend: materialise it first, then accumulate
                        self._add_into(gx, res, False)
                        acc = 1
                self.emit(N.OP_MAXPOOL_BWD, [dy.addr(), self.bp(am), gx.addr()],
                          [dy.ld, gx.ld, x.B, x.H, x.W, x.C, acc, self.dtype])
                self.grad_written(x)

            self.nodes.append(bwd)
        return y

    def global_avgpool(self, x: TRef, name="avgpool") -> TRef:
        """[B,H,W,C] -> [B,1,1,C]"""
        self.tag += 1
        y = self.act(x.B, 1, 1, x.C, name)
        self.emit(N.OP_AVGPOOL_FWD, [x.addr(), y.addr()], [x.ld, y.ld, x.B, x.H * x.W, x.C, self.dtype])
        if self.need_grad and x.needs_grad:
            tag = self.tag

            def bwd():
                self.tag = tag
                dy = self.grad_read(y)
                if dy is None:
                    return
                gx, res = self.grad_target(x)
                acc = 0
                if res is not None:
                    if not (res.buf is gx.buf and res.coff == gx.coff):
                        self._add_into(gx, res, False)
                    acc = 1
                self.emit(N.OP_AVGPOOL_BWD, [dy.addr(), gx.addr()],
                          [dy.ld, gx.ld, x.B, x.H * x.W, x.C, acc, self.dtype])
                self.grad_written(x)

            self.nodes.append(bwd)
        return y

    # -- ESE gate (reference vovnet.py:20-28) ---------------------------------------------
    def ese(self, x: TRef, linear: nn.Conv2d, residual: Optional[TRef] = None,
            out: Optional[TRef] = None, name="ese") -> TRef:
        pooled = self.global_avgpool(x, name + ".pool")
        s = self.conv_unit(pooled, linear, None, False, name=name + ".linear")
        self.tag += 1
        y = out if out is not None else self.act(x.B, x.H, x.W, x.C, name + ".y")
        self.emit(N.OP_ESE_FWD, [x.addr(), s.addr(), residual.addr() if residual else None, y.addr()],
                  [x.ld, s.ld, residual.ld if residual else 0, y.ld, x.B, x.H * x.W, x.C, self.dtype])
        if self.need_grad:
            tag = self.tag

            def bwd():
                self.tag = tag
                dy = self.grad_read(y)
                if dy is None:
                    return
                if residual is not None:
                    self.grad_add(residual, dy)
                gx, res = self.grad_target(x)
                acc = 0
                if res is not None:
                    if not (res.buf is gx.buf and res.coff == gx.coff):
                        self._add_into(gx, res, False)
                    acc = 1
                ds32 = self.f32(x.B * x.C, "ds32")
                self.emit(N.OP_ESE_BWD, [dy.addr(), x.addr(), s.addr(), gx.addr(), self.bp(ds32)],
                          [dy.ld, x.ld, s.ld, gx.ld, x.B, x.H * x.W, x.C, acc, self.dtype])
                self.grad_written(x)
                gs_, _ = self.grad_target(s)
                self.emit(N.OP_COPY2D, [self.bp(ds32), gs_.addr()], [N.VT_F32, self.dtype, x.C, 0],
                          [x.C, gs_.ld, x.B])

            self.nodes.append(bwd)
        return y

    # -- classifier head + loss (reference classifier.py:58-64, 92) -----------------------
    def xent(self, logits: TRef, label_smoothing: float, grad_scale: float, mix: bool = False) -> Buf:
        self.tag += 1
        loss = self.zeroed_f32(64, "loss")
        Bn, Ncls = logits.B, logits.C
        g = None
        if self.need_grad:
            gs = self._gs(logits)
            g = self._gref(logits)
            gs.init.append((logits.coff, logits.coff + logits.C))
        self.emit(N.OP_XENT, [logits.addr(), (LABELS, 0), self.bp(loss), g.addr() if g else None,
                              (HYPER, self.MIX_OFF) if mix else None],
                  [logits.ld, g.ld if g else 0, Bn, Ncls, self.dtype], [label_smoothing, grad_scale])
        return loss

    def xent_eval(self, logits: TRef) -> Buf:
        """validation (classifier.py:97-109): [loss sum without label smoothing, top-1 hits, rows] of the batch, f32[3]."""
        self.tag += 1
        out = self.zeroed_f32(64, "val_sums")
        self.emit(N.OP_XENT_EVAL, [logits.addr(), (LABELS, 0), self.bp(out)], [logits.ld, logits.B, logits.C, self.dtype])
        return out

    # -- finish ---------------------------------------------------------------------------
    def build_backward(self):
        self._cur = self.bwd
        for node in reversed(self.nodes):
            node()
        self._wgrad_release()  # (groups whose units did not all reach backward: a branch without gradient)
        self.emit(N.OP_JOIN)  # all filter gradients done before anything reads the gradient buffers
        self._cur = self.fwd
        if self._hoisted:
            # [FORK, packs on the side stream] ahead of the forward ops; the executor joins the
            # side stream at the end of the list, i.e. before the backward list starts
            # They are enqueued AFTER the first unit of the forward pass: issuing ~80 launches costs the
            # host ~0.5 ms, during which the main stream must already have work (measured: it sat idle
            # for exactly that long at the head of every step when the packs came first).
            body, self.fwd = self.fwd, []
            cut = next((i + 1 for i, op in enumerate(body) if (op.kind & 0xFFFF) in (N.OP_BN_ACT_APPLY, N.OP_BN_FIN_APPLY)), 0)
            self._cur = self.fwd
            self.fwd.extend(body[:cut])
            self.emit(N.OP_FORK)
            self.fwd.extend(self._hoisted)
            self.fwd.extend(body[cut:])
            self._hoisted = []

    def seed_output_grads(self, outs: list[TRef]):
        """reserve gradient buffers of the returned feature maps; they are filled from the
        caller's grad tensors before the backward list runs."""
        seeds = []
        for t in outs:
            g = self._gref(t)
            self._gs(t).init.append((t.coff, t.coff + t.C))
            seeds.append(g)
        return seeds


def ops_array(ops: list) -> "C.Array":
    arr = (N.Op * max(len(ops), 1))()
    for i, op in enumerate(ops):
        C.memmove(C.addressof(arr[i]), C.addressof(op), C.sizeof(N.Op))
    return arr


def tref_to_tensor(arena: torch.Tensor, t: TRef) -> torch.Tensor:
    """view of an arena activation as a logical-NCHW (channels_last strided) torch tensor."""
    # Built with set_() on the arena's storage rather than by slicing/viewing: the result shares
    # (and keeps alive) the arena memory but is NOT an autograd/tracer "view" of another tensor,
    # which is what autograd.Function outputs and torch.jit.trace need.
    td = _TORCH_DTYPE[t.dtype]
    byte_off = arena.storage_offset() + t.buf.offset + t.coff * t.esize
    assert byte_off % t.esize == 0
    out = torch.empty(0, dtype=td, device=arena.device)
    out.set_(arena.untyped_storage(), byte_off // t.esize, (t.B, t.C, t.H, t.W),
             (t.H * t.W * t.ld, 1, t.W * t.ld, t.ld))
    return out
