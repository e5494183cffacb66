"""Bucketed gradient all-reduce over RCCL (xGMI) for the data-parallel train step.

Replaces the DistributedDataParallel reducer that Lightning installs for the reference
(configs/base.yaml:17-19, `strategy: ddp_find_unused_parameters_false`): one process per
GPU, gradients live in ONE flat f32 buffer, and fixed contiguous slices of it ("buckets")
are all-reduced (sum; the 1/world average is folded into the SGD kernel).  Buckets are
issued in reverse parameter order -- the order backward produces them -- and
asynchronously: torch's NCCL(=RCCL) process group runs each collective on its own stream
after the work already enqueued on the caller's stream, so a bucket's transfer overlaps the
backward segments that follow it.

Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce of S
bytes costs ~2*(N-1)/N*S / link_bw, i.e. ~0.2 ms per 16 MiB on 8 GPUs, against a ~10 ms
step: 16 MiB buckets keep each transfer bandwidth-bound (>> the ~20 us launch latency)
while leaving only the small stem-side tail exposed after backward ends.

The same code runs on the `gloo` backend with CPU tensors (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist


def plan_buckets(total_elems: int, bucket_elems: int, align: int = 64, head_elems: int = 0) -> list[tuple[int, int]]:
    """contiguous [start, end) element ranges covering [0, total), LAST range first.

    Backward produces gradients from the end of the flat buffer towards its start, and the BatchNorm / bias gradients
    (the first region of the buffer: trainer.py orders the flat store norm | bias | everything else) are complete only
    when the stem's backward is: the bucket that starts at 0 is always the last one issued and its transfer is the
    exposed tail of the step.  `head_elems` > 0 makes that bucket exactly [0, head_elems) -- SURVEY 8(e): < 1 MiB, so
    the tail is latency only -- and cuts the rest into `bucket_elems` pieces from the end; the piece next to the head
    takes the remainder.  Every boundary is a multiple of `align` (for the sharded exchange: 64 * world)."""
    if total_elems <= 0:
        return []
    bucket_elems = max(align, bucket_elems // align * align)
    head = 0
    if head_elems > 0 and total_elems > head_elems:
        head = (head_elems + align - 1) // align * align
        if head >= total_elems:
            head = 0
    bounds = []
    end = total_elems
    while end > head:
        start = max(head, end - bucket_elems)
        bounds.append((start, end))
        end = start
    # merge a tiny remainder piece into its neighbour (never the head bucket)
    if len(bounds) >= 2 and bounds[-1][1] - bounds[-1][0] < bucket_elems // 4:
        s, _ = bounds.pop()
        s2, e2 = bounds.pop()
        bounds.append((s, e2))
    if head:
        bounds.append((0, head))
    return bounds


def _agree(ok: bool, group, device) -> bool:
    """True iff EVERY rank of `group` passed ok=True (one MIN all-reduce; gloo: host tensor, nccl: device tensor)."""
    backend = dist.get_backend(group)
    dev = device if (backend == "nccl" and device is not None) else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def ensure_library_comm(group=None, device: Optional[torch.device] = None, stat_comm: bool = False) -> int:
    """The library's own RCCL communicator over the ranks of `group` (vt_comm_init, include/vt_amd.h): rank 0 draws the
    id, torch.distributed carries its 128 bytes to the other ranks (any backend: this is the out-of-band channel the
    header speaks of, used once), every rank joins with its device current.  Returns the world size.  With it a launch
    list carries its collectives as ops (VT_OP_ALLREDUCE / VT_OP_STAT_SYNC): nothing of torch sits between two kernels.

    Failure is AGREED, never one-sided: every rank first runs the fallible local part (binding RCCL, drawing
    an id -- a purely local call) and the ranks all-reduce an ok flag BEFORE the id broadcast; a rank that cannot bind
    therefore raises on every rank instead of leaving the others blocked in the broadcast or in RCCL's bootstrap.  The
    collective join is followed by a second agreed flag.  `stat_comm`: also create the second communicator that
    vt_stat_sync uses (SyncBatchNorm exchanges must not queue behind bucket all-reduces of the same communicator).  Two
    communicators driven from two streams rely on both collectives being co-resident on the device (vt_comm.hip); the
    environment variable VT_STAT_COMM=0 keeps everything on the one communicator, whose issue order is the cross-rank order."""
    import ctypes
    import os

    if os.environ.get("VT_STAT_COMM", "1") == "0":
        stat_comm = False

    from . import _native as N

    L = N.lib()
    world = dist.get_world_size(group)
    have = L.vt_comm_world()
    if have and have != world:
        raise RuntimeError(f"the library communicator spans {have} ranks, the process group {world}")
    need_main, need_stat = not have, bool(stat_comm) and not L.vt_comm_has_stat()
    if not need_main and not need_stat:
        return world
    rank = dist.get_rank(group)
    src = dist.get_global_rank(group, 0) if group is not None else 0

    def join(init):
        # 1. the local, fallible part on EVERY rank; agree before anything collective depends on it
        ident = ctypes.create_string_buffer(128)
        err = ""
        try:
            N.check(L.vt_comm_unique_id(ident))
        except Exception as e:  # noqa: BLE001 -- reported on every rank below
            err = repr(e)
        if not _agree(not err, group, device):
            raise RuntimeError(f"library communicator: RCCL could not be bound on at least one rank (this rank: {err or 'ok'})")
        # 2. rank 0's id to everyone
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=src, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
        # 3. the collective join, then an agreed verdict
        err = ""
        try:
            ctx = torch.cuda.device(device) if device is not None and device.type == "cuda" else None
            if ctx is not None:
                with ctx:
                    N.check(init(ident))
            else:
                N.check(init(ident))
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        if not _agree(not err, group, device):
            raise RuntimeError(f"library communicator: the RCCL join failed on at least one rank (this rank: {err or 'ok'})")

    if need_main:
        join(lambda ident: L.vt_comm_init(ident, rank, world))
    if need_stat:
        join(lambda ident: L.vt_comm_init_stat(ident))
    return world


def self_test_library_comm(device: torch.device, timeout_s: float = 30.0, group=None) -> bool:
    """One small vt_allreduce_bucket on a stream of its own, polled from the host for at most `timeout_s`: True when it
    completed with the right sum.  A communicator that came up but whose first collective never finishes (a rank missing,
    a transport that does not connect) then costs one stuck 1 MiB kernel instead of the whole job -- the caller agrees on
    the verdict over torch.distributed and falls back (bench.py); the library communicator is NOT used after a False.
    `group`: the process group the communicator was built over (ensure_library_comm's) -- the rank that seeds the buffer is
    the rank INSIDE that group, which for a subgroup differs from the global one."""
    import time

    from . import _native as N

    L = N.lib()
    world, rank = L.vt_comm_world(), dist.get_rank(group)
    if not world:
        return False
    with torch.cuda.device(device):
        st = torch.cuda.Stream(device=device)
        buf = torch.full((262144,), float(rank + 1), dtype=torch.float32, device=device)
        torch.cuda.current_stream(device).synchronize()
        rc = L.vt_allreduce_bucket(buf.data_ptr(), buf.numel(), N.VT_F32, st.cuda_stream)
        if rc != N.VT_OK:
            return False
        t0 = time.perf_counter()
        while not st.query():
            if time.perf_counter() - t0 > timeout_s:
                return False
            time.sleep(0.005)
        want = world * (world + 1) / 2.0
        return bool((buf == want).all().item())


class GradBucketer:
    def __init__(self, flat: torch.Tensor, buckets: Sequence[tuple[int, int]], group=None):
        assert flat.dim() == 1 and flat.is_contiguous()
        covered = sorted(buckets)
        assert covered and covered[0][0] == 0 and covered[-1][1] == flat.numel()
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), "buckets must tile the buffer"
        self.flat, self.buckets, self.group = flat, list(buckets), group
        self.views = [flat[s:e] for s, e in self.buckets]
        self._pending: list = []

    def reduce_bucket(self, i: int) -> None:
        """start the all-reduce (sum) of bucket i; returns immediately."""
        work = dist.all_reduce(self.views[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(work)

    def reduce_all(self) -> None:
        for i in range(len(self.buckets)):
            self.reduce_bucket(i)

    def finish(self) -> None:
        """make the caller's stream (or thread, on gloo) wait for every started bucket."""
        for w in self._pending:
            w.wait()
        self._pending.clear()


class ShardedExchange:
    """reduce-scatter -> sharded optimiser -> all-gather, over the same buckets.

    Instead of all-reducing a bucket of the flat f32 gradient buffer (2 (N-1)/N S bytes per link and an optimiser
    step over every element on every rank), rank r receives the SUM of slice r of the bucket (reduce-scatter,
    (N-1)/N S), updates only that slice of parameters / momentum, and the updated weights travel back as an
    all-gather -- of the bf16 mirror the kernels read ((N-1)/N S/2) for conv / linear weights, of the f32 master values
    only for the head bucket, which holds the BatchNorm / bias parameters the kernels read in f32.  25 % fewer bytes on
    the links than the f32 all-reduce and 1/N of the optimiser traffic.  f32 master parameters and momentum of slices a
    rank does not own go stale on that rank: `gather_master()` refreshes them (checkpoints, state_dict).

    Every bucket boundary must be a multiple of 64 * world (plan_buckets(align=64 * world))."""

    def __init__(self, gflat: torch.Tensor, buckets: Sequence[tuple[int, int]], group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gflat, self.buckets = gflat, list(buckets)
        for s0, s1 in self.buckets:
            assert (s1 - s0) % self.world == 0 and s0 % 64 == 0, "bucket not divisible into equal 64-aligned shards"
        self.shards = [self.shard_of(i, self.rank) for i in range(len(self.buckets))]
        self._pending: list = []

    def shard_of(self, i: int, rank: int) -> tuple[int, int]:
        s0, s1 = self.buckets[i]
        n = (s1 - s0) // self.world
        return s0 + rank * n, s0 + (rank + 1) * n

    def reduce_bucket(self, i: int) -> None:
        """start the reduce-scatter (sum) of bucket i: this rank's slice of the bucket receives the total, in place"""
        s0, s1 = self.buckets[i]
        a, b = self.shards[i]
        self._pending.append(dist.reduce_scatter_tensor(self.gflat[a:b], self.gflat[s0:s1], op=dist.ReduceOp.SUM,
                                                        group=self.group, async_op=True))

    def reduce_all(self) -> None:
        for i in range(len(self.buckets)):
            self.reduce_bucket(i)

    def finish(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def gather(self, flat: torch.Tensor, which=None) -> None:
        """all-gather `flat` (same layout as the gradient buffer, any dtype) from the owners' slices, in place, for the
        buckets listed in `which` (default: all); blocks the caller's stream / thread until done"""
        works = []
        for i in (range(len(self.buckets)) if which is None else which):
            s0, s1 = self.buckets[i]
            a, b = self.shards[i]
            works.append(dist.all_gather_into_tensor(flat[s0:s1], flat[a:b], group=self.group, async_op=True))
        for w in works:
            w.wait()


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local, world
