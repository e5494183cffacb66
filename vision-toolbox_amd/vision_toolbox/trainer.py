"""Fused ImageNet-classifier train step on one MI355X, data-parallel over RCCL.

What the reference does through Lightning (classifier.py:58-64 model assembly, :83-95
training_step, :111-169 three-group SGD, configs/base.yaml:16-23 DDP) is here ONE static
program per rank:

    images -> backbone -> global avg-pool -> linear head -> label-smoothing CE
           -> explicit backward into a persistent flat f32 gradient buffer
           -> bucketed gradient all-reduce (RCCL over xGMI); each bucket is issued as soon
              as the backward segment that completes it has been enqueued
           -> SGD(momentum) over the flat buffers, which also refreshes the bf16 weights

The step is one launch list cut into segments at the bucket boundaries; every segment runs
through the native executor on two streams (main + filter-gradient side stream, left open
across segments so N > 1 keeps the N = 1 schedule; the bucket all-reduce is issued with the
side stream current).  hipGraph replay of the same segments is OPT-IN (use_graphs=True): it
measured 6 % slower than eager two-stream execution (21.2 against 20.0 ms) and one replay
ended in a host fault inside hipGraphLaunch that five re-runs did not reproduce (NOTEBOOK
R5.18, R6.1), so the default entry never goes through it.  BatchNorm uses per-rank batch statistics by
default; sync_bn=True gives the reference recipe's SyncBatchNorm (configs/base.yaml:22) at the
cost of 134 small collectives per step.

collectives="rccl" (round 4): the library's own RCCL communicator (vt_comm_init) instead of
torch.distributed -- the statistics exchanges and the bucket all-reduces are ops of the launch
lists (VT_OP_STAT_SYNC, VT_OP_ALLREDUCE), a step is three host calls for any world size, and a
SyncBatchNorm exchange stays on the stream of the kernels around it.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Optional

import torch
from torch import nn

from . import _native as N
from . import engine as E
from .distributed import GradBucketer, ShardedExchange, plan_buckets
from .program import Program, current_stream_handle

_NORMS = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.LayerNorm, nn.GroupNorm)
GROUP_NORM, GROUP_BIAS, GROUP_OTHER = 0, 1, 2  # flat-buffer order: small, late-produced groups first


class _LinearAsConv:
    """nn.Linear seen as a 1x1 convolution over a [B,1,1,C] map (classifier.py:63)."""

    kernel_size = (1, 1)
    stride = (1, 1)
    padding = (0, 0)
    dilation = (1, 1)
    groups = 1

    def __init__(self, linear: nn.Linear):
        self.weight, self.bias = linear.weight, linear.bias
        self.out_channels, self.in_channels = linear.out_features, linear.in_features


def param_groups(model: nn.Module) -> dict:
    """id(param) -> group, the split of classifier.py:111-155 (norm / bias / everything else)."""
    out = {}
    for mod in model.modules():
        for name, p in mod._parameters.items():
            if p is None:
                continue
            if isinstance(mod, _NORMS):
                g = GROUP_NORM
            elif isinstance(mod, (nn.Linear, nn.modules.conv._ConvNd)) and name == "bias":
                g = GROUP_BIAS
            else:
                g = GROUP_OTHER
            out[id(p)] = g
    return out


def warmup_cosine_lr(epoch: int, max_epochs: int, lr: float, warmup_epochs: int = 5, warmup_factor: float = 0.01,
                     decay_factor: float = 0.0) -> float:
    """LinearLR warm-up then CosineAnnealingLR, stepped per epoch (classifier.py:171-190)."""
    if warmup_epochs > 0 and epoch < warmup_epochs:
        return lr * (warmup_factor + (1 - warmup_factor) * epoch / warmup_epochs)
    t, T = epoch - warmup_epochs, max(max_epochs - warmup_epochs, 1)
    eta_min = lr * decay_factor
    return eta_min + (lr - eta_min) * (1 + math.cos(math.pi * t / T)) / 2


def sample_mix(cutmix_alpha: float, mixup_alpha: float, width: int, height: int):
    """Draw (mode, lambda, box) the way RandomCutMixMixUp.forward does (extras.py:96-109), with the
    same torch CPU RNG calls in the same order, so a seeded run pairs with the reference's:
    rand -> choose; then RandomMixup (extras.py:29,37) or RandomCutmix (extras.py:64,72-88)."""
    import math as _m

    if cutmix_alpha == 0 and mixup_alpha == 0:
        raise ValueError
    use_mixup = cutmix_alpha <= 0 or torch.rand(1).item() >= 0.5
    if use_mixup:
        if torch.rand(1).item() >= 1.0:  # p = 1: never taken, but the draw is part of the stream
            return "none", 1.0, (0, 0, 0, 0)
        lam = float(torch._sample_dirichlet(torch.tensor([mixup_alpha, mixup_alpha], dtype=torch.float32))[0])
        return "mixup", lam, (0, 0, 0, 0)
    if torch.rand(1).item() >= 1.0:
        return "none", 1.0, (0, 0, 0, 0)
    lam = float(torch._sample_dirichlet(torch.tensor([cutmix_alpha, cutmix_alpha], dtype=torch.float32))[0])
    r_x = int(torch.randint(width, (1,)))
    r_y = int(torch.randint(height, (1,)))
    r = 0.5 * _m.sqrt(1.0 - lam)
    r_w_half, r_h_half = int(r * width), int(r * height)
    x1, y1 = max(r_x - r_w_half, 0), max(r_y - r_h_half, 0)
    x2, y2 = min(r_x + r_w_half, width), min(r_y + r_h_half, height)
    lam = float(1.0 - (x2 - x1) * (y2 - y1) / (width * height))
    return "cutmix", lam, (x1, y1, x2, y2)


def reduce_validation_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """[loss sum, top-1 hits, rows] of this rank's batch -> the same three sums over all ranks of `group` (one scalar-sized
    all-reduce: the `sync_dist=True` of classifier.py:104; SURVEY collective C4).  Without a process group: unchanged."""
    if group is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return sums
    if torch.distributed.get_world_size(group) == 1:
        return sums
    t = sums.double()  # (hits and rows are integers: exact in any order)
    if torch.distributed.get_backend(group) != "nccl":
        t = t.cpu()
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t.to(sums.device, torch.float32)


def _grad_write_offsets(op: N.Op) -> list[int]:
    """element offsets in the flat gradient buffer this backward op writes."""
    return [op.ptr[k].offset // 4 for k in range(N.VT_OP_MAX_PTR) if op.ptr[k].base == E.GRADS]


class TrainStep:
    def __init__(
        self,
        backbone: nn.Module,
        num_classes: int = 1000,
        batch_size: int = 256,  # per rank
        image_size: int = 224,
        dtype: torch.dtype = torch.bfloat16,
        lr: float = 0.05,
        momentum: float = 0.9,
        weight_decay: float = 2e-5,
        norm_weight_decay: float = 0.0,
        bias_weight_decay: float = 0.0,
        label_smoothing: float = 0.1,
        device: Optional[torch.device] = None,
        process_group=None,
        bucket_mb: float = 16.0,
        use_graphs: bool = False,  # opt-in: slower than the two-stream executor, see the module docstring
        plan_only: bool = False,
        sync_bn: bool = False,
        mix: bool = False,
        freeze_bn: bool = False,
        deterministic: Optional[bool] = None,
        exchange: str = "allreduce",
        head_bucket_kb: float = 256.0,
        data_parallel: Optional[bool] = None,
        collectives: str = "torch",
    ):
        N.lib()
        self.device = torch.device(device if device is not None else "cuda")
        self.plan_only = plan_only  # build launch lists / bucket plan without a GPU (host-logic tests)
        if self.device.type != "cuda" and not plan_only:
            raise RuntimeError("TrainStep runs on the GPU only (no CPU fallback)")
        self.B, self.S = batch_size, image_size
        self.dtype = N.VT_BF16 if dtype == torch.bfloat16 else N.VT_F32
        self.pg = process_group
        self.world = 1
        # data_parallel=False: this rank's program alone, whatever process group exists (bench.py times the N = 1
        # denominator of a weak-scaling run on rank 0 of the same job)
        if data_parallel is not False and torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        # a one-rank process group still takes the data-parallel schedule (cut lists, bucket collectives on the
        # filter-gradient stream, broadcasts): how the RCCL path is exercised on a one-GPU box (tests)
        self.dp = data_parallel is not False and (
            self.world > 1 or (os.environ.get("VT_DP_WORLD1", "0") != "0" and torch.distributed.is_available()
                               and torch.distributed.is_initialized()))
        # (measurement only, set through the `exchange_skipped()` context manager, which restores the replicas' state)
        self._skip_exchange = False
        head = nn.Linear(backbone.get_last_out_channels(), num_classes)
        self.model = nn.Sequential(backbone, nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), head)
        self.model.train()
        if freeze_bn:
            # fine-tuning with frozen BatchNorm: every unit normalises with its running statistics (constants
            # in backward) and leaves them untouched, like `bn.eval()` inside a training model
            for m in self.model.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()

        groups = param_groups(self.model)
        self.store = st = E.ParamStore(self.model, order_key=lambda p: groups[id(p)])
        # gradient exchange: "allreduce" = one f32 all-reduce per bucket, every rank runs the whole optimiser;
        # "sharded" = reduce-scatter -> SGD on the rank's slice of every bucket -> all-gather of the bf16 weights
        if exchange not in ("allreduce", "sharded"):
            raise ValueError(exchange)
        self.exchange = exchange if self.dp else "allreduce"
        # who issues the collectives: "torch" = torch.distributed calls between segments of the launch lists (any
        # backend: gloo in the CPU tests); "rccl" = the library's own RCCL communicator (vt_comm_init), the collectives are
        # OPS of the lists (VT_OP_STAT_SYNC in front of every BatchNorm finalize, FORK + VT_OP_ALLREDUCE on the
        # filter-gradient stream behind the op that completes a bucket): a step is three host calls whatever the
        # number of ranks and layers, and no collective hops to another library's stream and back
        if collectives not in ("torch", "rccl"):
            raise ValueError(collectives)
        self.collectives = collectives if self.dp else "torch"
        if self.collectives == "rccl":
            if self.exchange != "allreduce":
                raise ValueError("collectives='rccl' carries the all-reduce exchange only")
            if not plan_only:
                from .distributed import ensure_library_comm

                # (with SyncBatchNorm: a second communicator for the statistics exchanges, see vt_comm_init_stat)
                ensure_library_comm(self.pg, self.device, stat_comm=bool(sync_bn) and self.dp)
        self._align = 64 * self.world if self.exchange == "sharded" else 64
        st.pad_multiple = self._align
        with self._dev_ctx():
            st.ensure(self.device)
            self.gflat = torch.zeros_like(st.pflat)
            self.mflat = torch.zeros_like(st.pflat)
            # HYPER buffer: [0] learning rate; [8..13] MixUp / CutMix block (mode, lambda, x1, y1, x2, y2)
            self.lr_dev = torch.zeros(16, dtype=torch.float32, device=self.device)
            self.lr_dev[:4] = lr
            self.images = torch.zeros(batch_size, 3, image_size, image_size, device=self.device)
            self.labels = torch.zeros(batch_size, dtype=torch.int64, device=self.device)
        self.lr = lr
        total = st.pflat.numel()  # (padded to the exchange's alignment; parameters end at st.total)

        # ---- forward + loss + backward launch lists ---------------------------------------
        # deterministic=True (default: the VT_DETERMINISTIC environment variable): order-free filter gradients and bias
        # sums on top of the always fixed-point BatchNorm statistics -- bit-identical parameter updates
        b = E.Builder(st, self.dtype, training=True, need_grad=True, grad_base=E.GRADS, deterministic=deterministic)
        self.deterministic = b.deterministic
        b.hoist_dgrad_packs = True
        # SyncBatchNorm, the reference recipe's setting (configs/base.yaml:22): batch statistics over
        # ALL ranks.  Off by default for the throughput metric: it adds two small,
        # strictly sequential collectives per unit (67 + 67 for CSPDarknet-53).
        self.sync_bn = bool(sync_bn) and self.dp
        b.bn_world = self.world if self.sync_bn else 1
        b.bn_sync = bool(self.sync_bn)
        self.mix = bool(mix)  # MixUp / CutMix applied on device from a per-step parameter block
        x = b.input_images(batch_size, 3, image_size, image_size, mix=self.mix)
        fmap = backbone._vt_emit_maps(b, x)[-1]
        pooled = b.global_avgpool(fmap, "head.pool")
        logits = b.conv_unit(pooled, _LinearAsConv(head), None, False, name="head.linear")
        self._loss_buf = b.xent(logits, label_smoothing, 1.0 / batch_size, mix=self.mix)
        self._logits = logits
        b.build_backward()
        self.prog = Program(b, [logits], [])
        self.n_units = b.n_units
        # sync points: the statistics a finalize kernel reads must be all-reduced right before it
        self._fwd_sync = self._sync_points(self.prog.fwd_ops, self.prog.n_fwd, N.OP_BN_FINALIZE) if self.sync_bn else []
        self._bwd_sync = self._sync_points(self.prog.bwd_ops, self.prog.n_bwd, N.OP_BN_BWD_FINALIZE) if self.sync_bn else []
        if self.collectives == "rccl" and self.sync_bn:
            # the statistics exchange becomes an op in front of each finalize kernel (same stream: ordered by position)
            self.prog.fwd_ops, self.prog.n_fwd = self._with_stat_sync(self.prog.fwd_ops, self.prog.n_fwd, self._fwd_sync)
            self.prog.bwd_ops, self.prog.n_bwd = self._with_stat_sync(self.prog.bwd_ops, self.prog.n_bwd, self._bwd_sync)
            self._fwd_sync, self._bwd_sync = [], []

        # ---- optimiser launch list: one SGD launch per weight-decay group ---------------------
        wd_of = {GROUP_OTHER: weight_decay, GROUP_NORM: norm_weight_decay, GROUP_BIAS: bias_weight_decay}
        self.segments = []  # (elem start, elem end, weight decay)
        gids = [groups[id(p)] for p in st.params]
        start = 0
        for i, p in enumerate(st.params):
            if i + 1 == len(st.params) or gids[i + 1] != gids[i]:
                end = st.offsets[i] + E._round_up(p.numel(), 64)
                self.segments.append((start, end, wd_of[gids[i]]))
                start = end
        self._momentum, self._lr0 = momentum, lr
        self._emit_opt([(0, total)])
        zb = E.Builder(st, self.dtype, True, True)
        zb.emit(N.OP_MEMSET, [(E.GRADS, 0)], [0], [total * 4])
        self.zero_ops = E.ops_array(zb.fwd)

        # ---- gradient buckets and the backward segments that complete them --------------------
        self.bucketer = None
        self.bwd_cuts = [self.prog.n_bwd]  # op index after which each segment ends
        self.cut_buckets: list[list[int]] = [[]]
        if self.dp:
            # the bucket that starts at 0 (BatchNorm / bias gradients + the stem-side filters) completes last: keep it
            # small, so that the exposed tail of the step is one latency-bound transfer (< 1 MiB)
            nb_end = next((o for o, g_ in zip(st.offsets, gids) if g_ == GROUP_OTHER), total)  # end of the norm | bias region
            head = max(int(head_bucket_kb * 1024) // 4, nb_end)
            buckets = plan_buckets(total, int(bucket_mb * (1 << 20)) // 4, align=self._align, head_elems=head)
            if self.exchange == "sharded":
                self.bucketer = ShardedExchange(self.gflat, buckets, self.pg)
                self._head_bucket = next((i for i, (b0, _) in enumerate(buckets) if b0 == 0), None)
                assert self._head_bucket is not None and buckets[self._head_bucket][1] >= nb_end, \
                    "the f32-read parameters (BatchNorm, biases) must sit in the head bucket"
                self._emit_opt(self.bucketer.shards)  # the optimiser touches this rank's slices only
            else:
                self.bucketer = GradBucketer(self.gflat, buckets, self.pg)
            ready = [0] * len(buckets)  # last bwd op that writes into each bucket
            # an op names the START of the gradient it writes; the gradient extends over the whole
            # parameter, which may straddle a bucket boundary -- every bucket the parameter
            # overlaps must wait for that op (missing this lets a tail bucket be reduced before
            # the op adds to it: ranks then apply different gradients and drift apart)
            import bisect

            starts = list(st.offsets)
            ends = [o + E._round_up(p.numel(), 64) for o, p in zip(st.offsets, st.params)]
            for idx in range(self.prog.n_bwd):
                for off in _grad_write_offsets(self.prog.bwd_ops[idx]):
                    k = bisect.bisect_right(starts, off) - 1
                    lo, hi = (starts[k], ends[k]) if k >= 0 and off < ends[k] else (off, off + 1)
                    for bi, (s0, s1) in enumerate(buckets):
                        if s0 < hi and lo < s1:
                            ready[bi] = max(ready[bi], idx + 1)
            cuts = sorted(set(ready) | {self.prog.n_bwd})
            self.bwd_cuts = [c for c in cuts if c > 0]
            self.cut_buckets = [[bi for bi, r in enumerate(ready) if r == c or (c == self.bwd_cuts[0] and r == 0)]
                                for c in self.bwd_cuts]
            if self.collectives == "rccl":
                self._inline_buckets(buckets)

        with self._dev_ctx():
            self.arena = torch.empty(1 if plan_only else self.prog.arena_bytes, dtype=torch.uint8, device=self.device)
            if self.dtype == N.VT_BF16:
                st.mirror.copy_(st.pflat)  # initial bf16 weights; afterwards SGD keeps them fresh
        self.bases = self.prog.bases(
            self.arena.data_ptr(), PARAMS=st.pflat.data_ptr(), GRADS=self.gflat.data_ptr(),
            STATE=st.sflat.data_ptr(), MIRROR=st.mirror.data_ptr(), COUNTERS=st.nflat.data_ptr(),
            INPUT=self.images.data_ptr(), LABELS=self.labels.data_ptr(), MOMENTUM=self.mflat.data_ptr(),
            HYPER=self.lr_dev.data_ptr())
        self.use_graphs = use_graphs
        self._master_stale = False
        self.model.register_state_dict_pre_hook(self._refuse_stale_export)
        self._side = None  # side stream for the filter gradients (eager mode; graphs fork internally)
        self._graphs = None
        self.steps_done = 0

    def _emit_opt(self, ranges):
        """optimiser launch list: one SGD launch per (element range, weight-decay segment) overlap"""
        ob = E.Builder(self.store, self.dtype, True, True, grad_base=E.GRADS)
        for r0, r1 in ranges:
            for s0, s1, wd in self.segments:
                lo, hi = max(r0, s0), min(r1, s1)
                if hi > lo:
                    ob.emit(N.OP_SGD,
                            [(E.PARAMS, lo * 4), (E.GRADS, lo * 4), (E.MOMENTUM, lo * 4),
                             (E.MIRROR, lo * 2) if self.dtype == N.VT_BF16 else None, (E.HYPER, 0)],
                            [N.VT_BF16], [hi - lo, self._lr0, self._momentum, wd, 1.0 / self.world])
        self.opt_ops, self.n_opt = E.ops_array(ob.fwd), len(ob.fwd)

    def _gather_weights(self) -> None:
        """sharded exchange: bring the updated weights of the other ranks' slices in -- the bf16 mirror the kernels
        read; f32 for the head bucket (BatchNorm / bias parameters are read in f32), whose mirror is re-cast here"""
        st, ex = self.store, self.bucketer
        if self.dtype == N.VT_BF16:
            rest = [i for i in range(len(ex.buckets)) if i != self._head_bucket]
            ex.gather(st.mirror, rest)
            ex.gather(st.pflat, [self._head_bucket])
            h0, h1 = ex.buckets[self._head_bucket]
            st.mirror[h0:h1].copy_(st.pflat[h0:h1])
        else:
            ex.gather(st.pflat)

    def gather_master(self) -> None:
        """sharded exchange: refresh the f32 master parameters and the momentum of the slices other ranks own (before a
        checkpoint / state_dict; a collective: every rank calls it); a no-op for the all-reduce exchange"""
        if self.exchange == "sharded":
            self.bucketer.gather(self.store.pflat)
            self.bucketer.gather(self.mflat)
        self._master_stale = False

    def _refuse_stale_export(self, *_):
        # state_dict() of the wrapped model reads views of the flat f32 buffer: after a sharded step the slices other
        # ranks own hold LAST step's values there, and a checkpoint written from them would silently mix old and new
        if self._master_stale:
            raise RuntimeError("exchange='sharded': the f32 master parameters of the slices other ranks own are stale "
                               "on this rank; call TrainStep.gather_master() on every rank before state_dict()")

    @staticmethod
    def _with_stat_sync(ops, n, points):
        """the launch list with a VT_OP_STAT_SYNC in front of every finalize op of `points` (on that op's stream)"""
        at = {idx for idx, *_ in points}
        out = []
        for idx in range(n):
            op = ops[idx]
            if idx in at:
                so = N.Op()
                so.kind = N.OP_STAT_SYNC | (op.kind & N.OP_SIDE_STREAM)
                so.tag = op.tag
                for k in range(N.VT_OP_MAX_PTR):
                    so.ptr[k].base = -1
                so.ptr[0].base, so.ptr[0].offset = op.ptr[0].base, op.ptr[0].offset
                so.i[0] = op.i[0]
                out.append(so)
            cp = N.Op()
            ctypes.memmove(ctypes.addressof(cp), ctypes.addressof(op), ctypes.sizeof(N.Op))
            out.append(cp)
        return E.ops_array(out), len(out)

    def _inline_buckets(self, buckets):
        """collectives='rccl': behind the op that completes a bucket, FORK (the filter-gradient stream waits for the main
        stream's position: BatchNorm / bias gradients are written there) and the bucket's all-reduce ON that stream --
        the main stream waits for nothing, the list's own JOIN closes the side stream before the optimiser.  The
        segments collapse into one list again."""
        p = self.prog
        after = dict(zip(self.bwd_cuts, self.cut_buckets))
        out = []

        def blank(kind):
            o = N.Op()
            o.kind = kind
            for k in range(N.VT_OP_MAX_PTR):
                o.ptr[k].base = -1
            return o

        open_side = False
        for idx in range(p.n_bwd):
            op = p.bwd_ops[idx]
            cp = N.Op()
            ctypes.memmove(ctypes.addressof(cp), ctypes.addressof(op), ctypes.sizeof(N.Op))
            out.append(cp)
            if (op.kind & 0xFFFF) == N.OP_JOIN:
                open_side = False
            bis = after.get(idx + 1, [])
            if bis:
                out.append(blank(N.OP_FORK))
                for bi in bis:
                    b0, b1 = buckets[bi]
                    o = blank(N.OP_ALLREDUCE | N.OP_SIDE_STREAM)
                    o.ptr[0].base, o.ptr[0].offset = E.GRADS, b0 * 4
                    o.i[0] = N.VT_F32
                    o.f[0] = float(b1 - b0)
                    out.append(o)
                open_side = True
        if open_side:  # (the head bucket completes with the last ops, behind the list's own JOIN)
            out.append(blank(N.OP_JOIN))
        self._bwd_without_exchange = (p.bwd_ops, p.n_bwd)  # (skip_exchange: the measurement of what the exchange exposes)
        p.bwd_ops, p.n_bwd = E.ops_array(out), len(out)
        self.bwd_cuts, self.cut_buckets = [p.n_bwd], [[]]
        self.bucketer = None
        self.inline_buckets = [buckets[bi] for bis in after.values() for bi in bis]

    @staticmethod
    def _sync_points(ops, n, kind):
        """[(op index of the finalize kernel, base id, byte offset, bytes of its [replicas][2][C] sums)]"""
        pts = []
        for idx in range(n):
            op = ops[idx]
            if (op.kind & 0xFFFF) == kind:
                pts.append((idx, op.ptr[0].base, op.ptr[0].offset, N.stat_floats(op.i[0]) * 4))
        return pts

    def _sync_view(self, base, off, nbytes):
        """the statistics of one BatchNorm as the all-reduce sees them: replica 0 of int64[replicas][2][C][2]"""
        start = {E.ZERO_F: self.prog.zf_off, E.ZERO_B: self.prog.zb_off}[base] + off
        # the sums are 64-bit fixed point (vt_amd.h, VT_STAT_REPLICAS): an integer all-reduce, exact and order-free
        return self.arena[start : start + nbytes // N.VT_STAT_REPLICAS].view(torch.int64)

    def _sync_stats(self, base, off, nbytes, stream):
        """SyncBatchNorm exchange of one layer's sums: the 32 replicas the kernels spread their atomics
        over are folded into replica 0 on the device first (vt_stat_fold), so the collective carries 32 C bytes --
        the algorithmic payload -- instead of the raw 1 KiB x C buffer"""
        start = {E.ZERO_F: self.prog.zf_off, E.ZERO_B: self.prog.zb_off}[base] + off
        C_ = nbytes // (N.VT_STAT_REPLICAS * 32)
        N.check(N.lib().vt_stat_fold(ctypes.c_void_p(self.arena.data_ptr() + start), C_, ctypes.c_void_p(stream)))
        torch.distributed.all_reduce(self._sync_view(base, off, nbytes), group=self.pg)

    def _run_list(self, ops, n, sync, cuts, cut_buckets, s, side, keep_side_open=True):
        """run a launch list in segments: a segment ends before every finalize kernel whose statistics
        must be all-reduced first (SyncBatchNorm) and after every op that completes a gradient bucket.

        Cutting the list must not cost the N = 1 schedule: an intermediate segment leaves the filter-gradient
        (side) stream open instead of joining it, and a bucket's collective is issued with the SIDE stream current,
        after ordering that stream behind the main stream's position (BatchNorm / bias gradients are written
        there).  RCCL's stream then waits for exactly the producers of the bucket, the main stream for nothing;
        the list's own JOIN op (last segment) and `bucketer.finish()` close both before the optimiser.
        `keep_side_open=False` (the forward list, which carries no JOIN op of its own): every segment joins the side
        stream on return, so the data-gradient filter packs hoisted onto it are ordered before backward reads them."""
        marks = {idx: ("sync", (base, off, nb)) for idx, base, off, nb in sync}
        ends = sorted(set(marks) | set(cuts) | {n})
        lo = 0
        for hi in ends:
            if hi > lo:
                sub = (N.Op * (hi - lo)).from_address(ctypes.addressof(ops) + lo * ctypes.sizeof(N.Op))
                N.run_ops(sub, hi - lo, self.bases, s, side=side, leave_side_open=keep_side_open and hi != n)
            if hi in marks:  # stream-ordered: NCCL makes the launch stream wait, no host sync
                self._sync_stats(*marks[hi][1], s)
            if hi in cut_buckets and cut_buckets[hi] and not self._skip_exchange:
                if side and self._side is not None:
                    N.stream_wait(side, s)
                    with torch.cuda.stream(self._side):
                        for bi in cut_buckets[hi]:
                            self.bucketer.reduce_bucket(bi)
                else:
                    for bi in cut_buckets[hi]:
                        self.bucketer.reduce_bucket(bi)
            lo = hi

    def _dev_ctx(self):
        import contextlib

        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    # -- data-parallel plumbing ----------------------------------------------------------------
    def exchange_skipped(self):
        """MEASUREMENT ONLY: a context in which step() runs the data-parallel schedule -- cut lists, stream ordering --
        WITHOUT issuing the collectives (step time with them minus step time without = what the exchange leaves exposed).
        Steps taken inside update every replica from its own rank's gradients, which would desynchronise the replicas (and
        under the sharded exchange leave non-owned slices stale): the context snapshots parameters, BatchNorm state, the
        batch counters, momentum and the bf16 mirror on entry and restores them on exit."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            st = self.store
            keep = [t.clone() for t in (st.pflat, st.sflat, st.nflat, self.mflat, st.mirror)]
            self._skip_exchange = True
            try:
                yield self
            finally:
                self._skip_exchange = False
                torch.cuda.synchronize(self.device)
                for dst, src_ in zip((st.pflat, st.sflat, st.nflat, self.mflat, st.mirror), keep):
                    dst.copy_(src_)

        return ctx()

    def broadcast_parameters(self, src: int = 0) -> None:
        """initial weights + buffers from rank `src` (what DDP's constructor does)."""
        if self.dp:
            # parameters, BatchNorm buffers (running statistics AND the int64 batch counters) and the optimiser's
            # momentum: a resumed run starts identical on every rank
            torch.distributed.broadcast(self.store.pflat, src, group=self.pg)
            torch.distributed.broadcast(self.store.sflat, src, group=self.pg)
            torch.distributed.broadcast(self.store.nflat, src, group=self.pg)
            torch.distributed.broadcast(self.mflat, src, group=self.pg)
            if self.dtype == N.VT_BF16:
                self.store.mirror.copy_(self.store.pflat)

    def weights_changed(self) -> None:
        """call after writing parameters from outside (load_state_dict, manual init)."""
        if self.dtype == N.VT_BF16:
            self.store.mirror.copy_(self.store.pflat)

    def set_lr(self, lr: float) -> None:
        self.lr = lr
        self.lr_dev[:4] = lr

    def set_mix(self, mode: str = "none", lam: float = 1.0, box=(0, 0, 0, 0)) -> None:
        """MixUp / CutMix parameters of the NEXT step(s) (TrainStep(mix=True)): mode 'none' | 'mixup' |
        'cutmix'; `lam` as sampled (mixup) or 1 - box area / image area (cutmix, extras.py:88);
        box = (x1, y1, x2, y2).  `sample_mix` draws them exactly as extras.py does."""
        if not self.mix:
            raise RuntimeError("this TrainStep was built without mix=True")
        code = {"none": 0.0, "mixup": 1.0, "cutmix": 2.0}[mode]
        vals = torch.tensor([code, float(lam), *[float(v) for v in box], 0.0, 0.0], dtype=torch.float32)
        self.lr_dev[8:16].copy_(vals, non_blocking=True)

    # -- one step ----------------------------------------------------------------------------------
    def _segment_ops(self):
        """(ops pointer, count) of every backward segment."""
        p = self.prog
        segs, lo = [], 0
        for hi in self.bwd_cuts:
            sub = (N.Op * (hi - lo)).from_address(ctypes.addressof(p.bwd_ops) + lo * ctypes.sizeof(N.Op))
            segs.append((sub, hi - lo))
            lo = hi
        return segs

    def _build_graphs(self):
        p = self.prog
        with self._dev_ctx():
            self._graphs = {
                "head": N.Graph(E.ops_array(list(self.zero_ops[:1]) + [p.fwd_ops[i] for i in range(p.n_fwd)]),
                                1 + p.n_fwd, self.bases),
                "bwd": [N.Graph(ops, n, self.bases) for ops, n in self._segment_ops()],
                "opt": N.Graph(self.opt_ops, self.n_opt, self.bases),
            }

    def step(self, images: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None) -> None:
        """fwd + loss + bwd + gradient all-reduce + SGD on the resident (or given) batch."""
        if self.plan_only:
            raise RuntimeError("plan_only TrainStep cannot execute: there is no CPU path")
        with self._dev_ctx():
            if images is not None:
                self.images.copy_(images, non_blocking=True)
            if labels is not None:
                self.labels.copy_(labels, non_blocking=True)
            s = current_stream_handle()
            p = self.prog
            if self.use_graphs and not self.sync_bn and self.collectives != "rccl":
                if self._graphs is None:
                    self._build_graphs()
                self._graphs["head"].launch(s)
                for g, bks in zip(self._graphs["bwd"], self.cut_buckets):
                    g.launch(s)
                    for bi in ([] if self._skip_exchange else bks):
                        self.bucketer.reduce_bucket(bi)
            else:
                if self._side is None:
                    self._side = torch.cuda.Stream(self.device)
                side = int(self._side.cuda_stream)
                N.run_ops(self.zero_ops, 1, self.bases, s)
                self._run_list(p.fwd_ops, p.n_fwd, self._fwd_sync, [], {}, s, side, keep_side_open=False)
                bwd_ops, n_bwd = p.bwd_ops, p.n_bwd
                if self._skip_exchange and self.collectives == "rccl" and getattr(self, "_bwd_without_exchange", None):
                    bwd_ops, n_bwd = self._bwd_without_exchange
                self._run_list(bwd_ops, n_bwd, self._bwd_sync, [n_bwd] if self.collectives == "rccl" else self.bwd_cuts,
                               dict(zip(self.bwd_cuts, self.cut_buckets)), s, side)
            if self.bucketer is not None and not self._skip_exchange:
                self.bucketer.finish()
            if self.use_graphs and not self.sync_bn and self.collectives != "rccl":
                self._graphs["opt"].launch(s)
            else:
                N.run_ops(self.opt_ops, self.n_opt, self.bases, s)
            if self.exchange == "sharded" and not self._skip_exchange:
                self._gather_weights()
                self._master_stale = True
        self.steps_done += 1

    # ---- validation step (reference classifier.py:97-109) ----------------------------------------------------------
    def _build_eval(self):
        """The SAME modules and flat parameter store compiled once more in eval mode: every BatchNorm uses its running
        statistics (folded into the conv epilogues: one launch per unit), no gradient bookkeeping, and the loss op is the
        validation one (no label smoothing, top-1 hits).  Its arena is its own (forward only)."""
        was = [(m, m.training) for m in self.model.modules()]
        self.model.eval()
        try:
            b = E.Builder(self.store, self.dtype, training=False, need_grad=False)
            x = b.input_images(self.B, 3, self.S, self.S)
            fmap = self.model[0]._vt_emit_maps(b, x)[-1]
            pooled = b.global_avgpool(fmap, "head.pool")
            logits = b.conv_unit(pooled, _LinearAsConv(self.model[3]), None, False, name="head.linear")
            sums = b.xent_eval(logits)
            b.build_backward()
            prog = Program(b, [logits], [])
        finally:
            for m, t in was:
                m.training = t
        with self._dev_ctx():
            arena = torch.empty(prog.arena_bytes, dtype=torch.uint8, device=self.device)
        bases = prog.bases(arena.data_ptr(), PARAMS=self.store.pflat.data_ptr(), STATE=self.store.sflat.data_ptr(),
                           MIRROR=self.store.mirror.data_ptr(), INPUT=self.images.data_ptr(),
                           COUNTERS=self.store.nflat.data_ptr(), LABELS=self.labels.data_ptr())
        self._eval = (prog, arena, bases, prog.zf_off + sums.offset, logits)

    def validate(self, images: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None) -> dict:
        """One validation step on the resident (or given) batch: eval-mode forward, cross entropy WITHOUT label smoothing,
        top-1 accuracy -- `validation_step` of classifier.py:97-109.  Data parallel, the three sums (loss, hits, rows) are
        all-reduced over the ranks (its `sync_dist=True`; collective C4 of SURVEY section 2), so every rank returns the
        GLOBAL mean loss and accuracy.  Parameters, BatchNorm statistics and optimiser state are not touched."""
        if self.plan_only:
            raise RuntimeError("plan_only TrainStep cannot execute: there is no CPU path")
        if getattr(self, "_master_stale", False):
            self.gather_master()
        with self._dev_ctx():
            if images is not None:
                self.images.copy_(images, non_blocking=True)
            if labels is not None:
                self.labels.copy_(labels, non_blocking=True)
            if getattr(self, "_eval", None) is None:
                self._build_eval()
            prog, arena, bases, off, _ = self._eval
            s = current_stream_handle()
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            N.run_ops(prog.fwd_ops, prog.n_fwd, bases, s, side=int(self._side.cuda_stream))
            sums = arena[off : off + 12].view(torch.float32).clone()
        sums = reduce_validation_sums(sums, self.pg if self.dp else None)
        loss_sum, hits, rows = (float(v) for v in sums.tolist())
        return {"loss": loss_sum / rows, "acc": hits / rows, "correct": int(round(hits)), "count": int(round(rows))}

    def eval_logits(self) -> torch.Tensor:
        """logits of the last validate() call"""
        _, arena, _, _, logits = self._eval
        return E.tref_to_tensor(arena, logits).reshape(self.B, -1)

    def loss(self) -> float:
        """mean loss of the last step (synchronises)."""
        off = self.prog.zf_off + self._loss_buf.offset
        return float(self.arena[off : off + 4].view(torch.float32).item())

    def logits(self) -> torch.Tensor:
        return E.tref_to_tensor(self.arena, self._logits).reshape(self.B, -1)
