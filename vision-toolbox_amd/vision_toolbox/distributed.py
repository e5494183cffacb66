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


def plan_buckets(total_elems: int, bucket_elems: int, align: int = 64) -> list[tuple[int, int]]:
    """contiguous [start, end) element ranges covering [0, total), LAST range first.

    The first (stem-side) bucket absorbs the remainder so the final, exposed transfer is the
    smallest one."""
    if total_elems <= 0:
        return []
    bucket_elems = max(align, bucket_elems // align * align)
    bounds = []
    end = total_elems
    while end > 0:
        start = max(0, end - bucket_elems)
        bounds.append((start, end))
        end = start
    # merge a tiny head bucket into its neighbour
    if len(bounds) >= 2 and bounds[-1][1] - bounds[-1][0] < bucket_elems // 4:
        s, _ = bounds.pop()
        s2, e2 = bounds.pop()
        bounds.append((s, e2))
    return bounds


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
