"""Data-parallel path on CPU: world_size-2 `gloo` processes exercise the bucket plan, the
bucketed gradient all-reduce and the parameter broadcast (the N>1 path of bench.py, which
uses the same code on RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_toolbox.distributed import GradBucketer, plan_buckets


def test_plan_buckets_tiles_the_buffer_last_range_first():
    for total, bucket in [(27_269_632, 4 << 20), (1000, 64), (64, 4096), (130, 64), (8_388_608, 4 << 20)]:
        b = plan_buckets(total, bucket)
        assert b[0][1] == total and sorted(b)[0][0] == 0
        s = sorted(b)
        assert all(x[1] == y[0] for x, y in zip(s, s[1:]))
        assert b == sorted(b, reverse=True)  # issued from the tail (produced first) to the head
        assert all(e - s_ <= 2 * max(bucket, 64) for s_, e in b)
    assert plan_buckets(0, 64) == []


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root / "vision-toolbox_amd"), str(root)]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        n = 10_000
        flat = torch.randn(n)
        mine = flat.clone()
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, mine)
        gb = GradBucketer(flat, plan_buckets(n, 2048))
        assert len(gb.buckets) >= 4
        gb.reduce_all()
        gb.finish()
        ok_sum = torch.allclose(flat, sum(gathered), rtol=1e-6, atol=1e-6)

        # the train step's bucket <-> backward-segment plan and the initial broadcast
        from vision_toolbox import backbones
        from vision_toolbox.trainer import TrainStep

        torch.manual_seed(rank)  # different init per rank on purpose
        ts = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True,
                       bucket_mb=0.5)
        assert ts.world == world and ts.bucketer is not None
        issued = [bi for group in ts.cut_buckets for bi in group]
        ok_plan = sorted(issued) == list(range(len(ts.bucketer.buckets))) and ts.bwd_cuts[-1] == ts.prog.n_bwd \
            and ts.bwd_cuts == sorted(ts.bwd_cuts) and len(ts.bwd_cuts) == len(ts.cut_buckets)
        # the tail bucket (head / last stage weights) must be ready no later than the stem-side one
        first_seg_of = {bi: si for si, grp in enumerate(ts.cut_buckets) for bi in grp}
        ok_order = first_seg_of[0] <= first_seg_of[len(ts.bucketer.buckets) - 1]
        # a bucket may only be reduced after EVERY op that writes any part of it: parameters straddle
        # bucket boundaries, so the op that names a gradient's start also owns its tail in the next bucket
        from vision_toolbox import engine as E
        from vision_toolbox.trainer import _grad_write_offsets

        seg_end_of = {bi: ts.bwd_cuts[si] for si, grp in enumerate(ts.cut_buckets) for bi in grp}
        extent = {o: o + p.numel() for o, p in zip(ts.store.offsets, ts.store.params)}
        ok_ready, straddlers = True, 0
        for idx in range(ts.prog.n_bwd):
            for off in _grad_write_offsets(ts.prog.bwd_ops[idx]):
                lo, hi = off, extent.get(off, off + 1)
                hit = [bi for bi, (s0, s1) in enumerate(ts.bucketer.buckets) if s0 < hi and lo < s1]
                straddlers += len(hit) > 1
                ok_ready &= all(seg_end_of[bi] >= idx + 1 for bi in hit)
        ok_plan = ok_plan and ok_ready and straddlers > 0  # the case must actually occur in this plan
        ts.broadcast_parameters(0)
        ref = ts.store.pflat.clone()
        dist.broadcast(ref, 0)
        ok_bcast = torch.equal(ref, ts.store.pflat)
        # averaging is folded into SGD: grad_scale = 1 / world
        ok_scale = abs(ts.opt_ops[0].f[4] - 1.0 / world) < 1e-12
        q.put((rank, ok_sum, ok_plan, ok_order, ok_bcast, ok_scale, _launch_list_digest(ts)))
    finally:
        dist.destroy_process_group()


def _launch_list_digest(ts):
    """digest of the forward and backward launch lists (every op: kind, operands, integer / float arguments)"""
    import ctypes
    import hashlib

    from vision_toolbox import _native as N

    h = hashlib.sha256()
    for ops, n in ((ts.prog.fwd_ops, ts.prog.n_fwd), (ts.prog.bwd_ops, ts.prog.n_bwd)):
        h.update(ctypes.string_at(ctypes.addressof(ops), n * ctypes.sizeof(N.Op)))
    return h.hexdigest()


def test_two_rank_gloo_bucketed_allreduce_and_train_plan():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    for r in results:
        assert all(r[1:]), r
    # a rank of the data-parallel job runs EXACTLY the single-GPU launch lists -- the collectives
    # are issued between segments of the same list (and from the filter-gradient stream), nothing is re-planned
    from vision_toolbox import backbones
    from vision_toolbox.trainer import TrainStep

    solo = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True, bucket_mb=0.5)
    assert solo.world == 1 and solo.bucketer is None
    assert {r[6] for r in results} == {_launch_list_digest(solo)}


def test_bench_launches_its_own_ranks_from_one_command():
    """`python bench.py --gpus 2` (the driver's command shape) must start its two ranks itself, the way
    Lightning does for the reference (configs/base.yaml:17-19).  Plan-only over gloo: launch lists and the
    bucket plan are built on the CPU, then the first collectives (parameter broadcast, every gradient bucket)
    run; rank 0 prints one JSON line."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, VT_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--plan-only", "--model",
                          "darknet_yolov5n", "--batch", "2", "--image-size", "64", "--bucket-mb", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["allreduce_ok"] and rec["params_equal"] and rec["buckets"] >= 2
    # the N > 1 line is self-contained (round 4): its own same-batch N = 1 denominator, the efficiency against it and
    # the exposed part of the gradient exchange -- timed on a GPU, null in a plan-only run, whose rank 0 still builds
    # the single-GPU plan beside the data-parallel one and finds the same launches in it
    for key in ("n1_same_per_gpu_batch_ms", "n1_same_per_gpu_batch_images_per_sec", "weak_scaling_efficiency",
                "exchange_exposed_ms"):
        assert key in rec
    assert rec["n1_plan"]["bwd_segments"] == 1 and rec["n1_plan"]["launches"] > 100
    assert rec["bwd_segments"] >= 2


def test_plan_buckets_head_bucket_and_alignment():
    """the bucket that starts at 0 (BatchNorm / bias gradients, complete only when the stem's backward is)
    is issued last and is the exposed tail of the step: it is exactly the requested head, and every boundary honours the
    alignment the sharded exchange needs (64 * world)"""
    for total, bucket, align, head in [(27_269_632, 4 << 20, 512, 65_536), (1 << 20, 1 << 16, 256, 40_000), (4096, 1024, 64, 100)]:
        b = plan_buckets(total, bucket, align=align, head_elems=head)
        s = sorted(b)
        assert s[0][0] == 0 and s[-1][1] == total and all(x[1] == y[0] for x, y in zip(s, s[1:]))
        assert b == sorted(b, reverse=True) and b[-1][0] == 0
        assert all(v % align == 0 for pair in b for v in pair)
        assert head <= b[-1][1] < head + align  # the head bucket: the requested size, rounded up to the alignment
        assert all(e - s_ <= 2 * bucket for s_, e in b[:-1])
    assert plan_buckets(64, 4096, head_elems=1000) == [(0, 64)]  # a head that swallows the buffer: one bucket


def _cpu_sgd(ts):
    """what the optimiser launch list does (vt_sgd_momentum: g' = g*grad_scale + wd*p; m = mu*m + g'; p -= lr*m;
    mirror = bf16(p)), interpreted on CPU tensors"""
    for k in range(ts.n_opt):
        op = ts.opt_ops[k]
        n, lr, mu, wd, gs = int(op.f[0]), *[torch.tensor(op.f[i], dtype=torch.float32) for i in (1, 2, 3, 4)]
        lo = op.ptr[0].offset // 4
        p, g, m = ts.store.pflat[lo:lo + n], ts.gflat[lo:lo + n], ts.mflat[lo:lo + n]
        m.mul_(mu).add_(g * gs + wd * p)
        p.sub_(lr * m)
        ts.store.mirror[lo:lo + n].copy_(p)


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root / "vision-toolbox_amd"), str(root)]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vision_toolbox import backbones
        from vision_toolbox.trainer import TrainStep

        steps = {}
        for mode in ("allreduce", "sharded"):
            torch.manual_seed(7)
            ts = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True,
                           bucket_mb=0.5, exchange=mode, weight_decay=1e-3, lr=0.1)
            ts.broadcast_parameters(0)
            steps[mode] = ts
        a, b = steps["allreduce"], steps["sharded"]
        assert b.exchange == "sharded" and b.store.pflat.numel() % (64 * world) == 0
        n = a.store.total
        ok = torch.equal(a.store.pflat[:n], b.store.pflat[:n])
        head = b.bucketer.buckets[b._head_bucket]
        ok_head = head[0] == 0 and (head[1] - head[0]) * 4 <= (1 << 20) and b._head_bucket in [g_ for g_ in b.cut_buckets if g_][-1]
        # the sharded optimiser touches exactly this rank's slices
        touched = sorted((op.ptr[0].offset // 4, op.ptr[0].offset // 4 + int(op.f[0])) for op in (b.opt_ops[k] for k in range(b.n_opt)))
        ok_own = all(any(s0 <= lo and hi <= s1 for s0, s1 in b.bucketer.shards) for lo, hi in touched) and \
            sum(hi - lo for lo, hi in touched) <= sum(s1 - s0 for s0, s1 in b.bucketer.shards)
        g = torch.Generator().manual_seed(1000 + rank)
        for step in range(2):  # (two steps: the momentum of the first one takes part)
            # integer-valued gradients (exact in f32 whatever the order of the ranks' contributions)
            grad = torch.randint(-8, 9, (n,), generator=g).float()
            for ts in (a, b):
                ts.gflat.zero_()
                ts.gflat[:n] = grad
                ts.bucketer.reduce_all()
                ts.bucketer.finish()
                _cpu_sgd(ts)
            b._gather_weights()
            # what the kernels read next step is identical on both paths, bit for bit: bf16 weights everywhere, f32
            # BatchNorm / bias parameters (head bucket)
            ok &= torch.equal(a.store.mirror[:n].view(torch.int16), b.store.mirror[:n].view(torch.int16))
            ok &= torch.equal(a.store.pflat[:head[1]], b.store.pflat[:head[1]])
            for s0, s1 in b.bucketer.shards:  # the owner's f32 master values and momentum
                s1 = min(s1, n)
                ok &= torch.equal(a.store.pflat[s0:s1], b.store.pflat[s0:s1]) and torch.equal(a.mflat[s0:s1], b.mflat[s0:s1])
        b.gather_master()
        ok &= torch.equal(a.store.pflat[:n], b.store.pflat[:n]) and torch.equal(a.mflat[:n], b.mflat[:n])
        q.put((rank, bool(ok), bool(ok_head), bool(ok_own)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_exchange_equals_the_allreduce_path_bit_for_bit(world):
    """reduce-scatter -> sharded SGD -> bf16 all-gather (TrainStep(exchange="sharded")) against the f32
    all-reduce path on the same gradients: the weights the next step reads are bit-identical on every rank"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for r in results:
        assert all(r[1:]), r


def test_inlist_collectives_layout_of_the_launch_lists(monkeypatch):
    """collectives="rccl" (round 4; include/vt_amd.h vt_allreduce_bucket / vt_stat_sync): the data-parallel collectives are
    ops OF the launch lists.  Host logic only (plan_only, a one-rank gloo group stands in for the job): every gradient
    bucket is one VT_OP_ALLREDUCE on the filter-gradient stream behind a FORK, the buckets tile the flat gradient buffer
    exactly once, a JOIN follows the last of them, no op that writes into a bucket comes after its all-reduce, and with
    sync_bn=True a VT_OP_STAT_SYNC on the finalize kernel's own sums sits directly in front of every BatchNorm finalize
    (forward and backward) -- the reference's `sync_batchnorm: true` / DDP reducer (configs/base.yaml:17-22)."""
    from vision_toolbox import _native as N
    from vision_toolbox import backbones
    from vision_toolbox import engine as E
    from vision_toolbox.trainer import TrainStep, _grad_write_offsets

    monkeypatch.setenv("VT_DP_WORLD1", "1")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        ts = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True, bucket_mb=0.5,
                       sync_bn=True, collectives="rccl")
        plain = TrainStep(backbones.darknet_yolov5n(), 16, 2, 64, torch.bfloat16, device="cpu", plan_only=True,
                          data_parallel=False)
    finally:
        dist.destroy_process_group()
    assert ts.collectives == "rccl" and ts.bucketer is None and ts.bwd_cuts == [ts.prog.n_bwd]
    for ops, n, fin, n_plain in ((ts.prog.fwd_ops, ts.prog.n_fwd, N.OP_BN_FINALIZE, plain.prog.n_fwd),
                                 (ts.prog.bwd_ops, ts.prog.n_bwd, N.OP_BN_BWD_FINALIZE, plain.prog.n_bwd)):
        kinds = [ops[i].kind & 0xFFFF for i in range(n)]
        nfin = kinds.count(fin)
        assert nfin > 10 and kinds.count(N.OP_STAT_SYNC) == nfin
        for i, k in enumerate(kinds):
            if k == fin:
                s, f = ops[i - 1], ops[i]
                assert (s.kind & 0xFFFF) == N.OP_STAT_SYNC and (s.kind & N.OP_SIDE_STREAM) == (f.kind & N.OP_SIDE_STREAM)
                assert (s.ptr[0].base, s.ptr[0].offset, s.i[0]) == (f.ptr[0].base, f.ptr[0].offset, f.i[0])
        # nothing else was added or lost against the single-GPU lists (whose BatchNorm finalize steps sit inside the
        # normalise / apply passes: with SyncBatchNorm they are launches of their own again, behind the exchange)
        skip = (N.OP_STAT_SYNC, N.OP_ALLREDUCE, N.OP_FORK, N.OP_JOIN, N.OP_BN_FINALIZE, N.OP_BN_BWD_FINALIZE)
        pops = plain.prog.fwd_ops if fin == N.OP_BN_FINALIZE else plain.prog.bwd_ops
        assert sum(k not in skip for k in kinds) == sum((pops[i].kind & 0xFFFF) not in skip for i in range(n_plain))
    ops, n = ts.prog.bwd_ops, ts.prog.n_bwd
    covered = []
    last_ar = -1
    for i in range(n):
        if (ops[i].kind & 0xFFFF) == N.OP_ALLREDUCE:
            assert ops[i].kind & N.OP_SIDE_STREAM and ops[i].ptr[0].base == E.GRADS and ops[i].i[0] == N.VT_F32
            j = i - 1
            while (ops[j].kind & 0xFFFF) == N.OP_ALLREDUCE:
                j -= 1
            assert (ops[j].kind & 0xFFFF) == N.OP_FORK  # the side stream is ordered behind the bucket's producers
            lo = ops[i].ptr[0].offset // 4
            hi = lo + int(ops[i].f[0])
            covered.append((lo, hi))
            last_ar = i
            for k in range(i + 1, n):  # nothing writes into the bucket afterwards
                for off in _grad_write_offsets(ops[k]):
                    assert not (lo <= off < hi), (i, k, off)
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == ts.gflat.numel() and len(covered) > 2
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert any((ops[i].kind & 0xFFFF) == N.OP_JOIN for i in range(last_ar + 1, n))


def _val_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root / "vision-toolbox_amd"), str(root)]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vision_toolbox.trainer import reduce_validation_sums

        mine = torch.tensor([1.5 * (rank + 1), float(3 + rank), 8.0])
        q.put((rank, reduce_validation_sums(mine).tolist()))
    finally:
        dist.destroy_process_group()


def test_validation_sums_are_all_reduced_over_the_ranks():
    """the scalar-sized exchange of the validation step (`sync_dist=True`, classifier.py:104; SURVEY collective C4)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] == [4.5, 7.0, 16.0]


def test_validation_sums_without_a_process_group_are_unchanged():
    from vision_toolbox.trainer import reduce_validation_sums

    t = torch.tensor([2.0, 1.0, 4.0])
    assert torch.equal(reduce_validation_sums(t), t)
