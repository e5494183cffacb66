#!/usr/bin/env python
"""Headline benchmark: images/sec of one CSPDarknet-53 bf16 train step @224px.

    python bench.py --gpus N --steps K --warmup W

A step = forward + label-smoothing CE + backward + gradient all-reduce (N>1) + SGD(momentum)
over one synthetic batch resident in HBM: 3x224x224, uniform [0,1) images, random-init weights.
N = 1 runs BASELINE.json configs[1] (batch 256 on one GPU); N > 1 runs configs[2] (data parallel, 128 images
per GPU = global 128*N, 1024 on 8 GPUs: the reference recipe's global batch, data.py:65-66) -- weak scaling at
128 per GPU, whose 1-GPU denominator the N = 1 line carries as `n1_same_per_gpu_batch_ms`.  `--batch` overrides.
Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel
(measured live, IN SITU: HIP events around that layer's launches inside the step's own forward
list, on the launch stream) and the CPU baseline (the oracle, timed on the host cores on a
bounded sample).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N
ranks (fresh child processes, one per GPU, rendezvous on 127.0.0.1) the way Lightning spawns the
reference's ranks from one command (configs/base.yaml:17-19); under torch.distributed.run it
uses the ranks it is given.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT / "vision-toolbox_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LABELS = {"cspdarknet53": "CSPDarknet-53", "darknet53": "Darknet-53", "darknet19": "Darknet-19", "vovnet39": "VoVNet-39"}
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def conv_roofline(B, C, HW, dtype_id, iters=30, warmup=10, k=3, Cout=None):
    """time one conv kernel standalone (HIP events on its launch stream) on a layer shape of the
    model: C -> Cout (default C) kxk stride 1 at HWxHW, batch B, with the training epilogue (BN
    statistics).  The dominant kernel is the input-span implicit-GEMM 3x3 conv."""
    from vision_toolbox import _native as N

    Cout = Cout or C
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn(B, HW, HW, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, C, device=dev) * (2.0 / (k * k * C)) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, HW, HW, Cout, device=dev, dtype=torch.bfloat16)
    stats = N.stats_buffer(Cout)
    d = N.ConvDesc()
    d.dtype = dtype_id
    d.B, d.Hi, d.Wi, d.Cin, d.ldx = B, HW, HW, C, C
    d.Ho, d.Wo, d.sh, d.sw, d.h0, d.w0 = HW, HW, 1, 1, -(k // 2), -(k // 2)
    d.Cout, d.ldy, d.oH, d.oW, d.oHs, d.oWs = Cout, Cout, HW, HW, 1, 1
    d.ldw, d.flags, d.ntaps = k * k * C, N.VT_CONV_STATS, k * k
    for i in range(k * k):
        d.dh[i], d.dw[i] = i // k, i % k
    s = int(torch.cuda.current_stream().cuda_stream)
    lib = N.lib()

    def launch():
        N.check(lib.vt_conv_igemm(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None,
                                  stats.data_ptr(), s))

    for _ in range(warmup):
        launch()
    e0, e1 = N.Event(), N.Event()
    e0.record(s)
    for _ in range(iters):
        launch()
    e1.record(s)
    ms = e0.elapsed_ms(e1) / iters
    flops = 2.0 * B * HW * HW * Cout * k * k * C
    nbytes = 2.0 * (B * HW * HW * (C + Cout) + Cout * k * k * C)  # input + output + filter once, bf16
    return {"ms": ms, "tflops": flops / ms / 1e9, "flops": flops, "gbs": nbytes / ms / 1e6, "bytes": nbytes,
            "shape": f"conv{k}x{k} s1 {C}->{Cout} @{HW}x{HW} B={B} (M={B*HW*HW} N={Cout} K={k*k*C})"}


def self_launch(n: int) -> int:
    """start `n` ranks of this script as child processes (never re-exec a process that touched the
    GPU: the parent has not, and only waits); rank 0 prints the JSON line; returns the worst rc."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a dead rank leaves the others stuck in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def host_cpu():
    """(physical cores, model name) of the host; os.cpu_count() counts SMT threads"""
    model, pairs, phys, core = "", set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and not model:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                pairs.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
    except OSError:
        pass
    ncores = len(pairs) or (os.cpu_count() or 1)
    try:  # respect a cgroup / affinity limit (the build container exposes 8 CPUs of a bigger host)
        ncores = min(ncores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return ncores, model


def insitu_layer_times(ts, Cin, Cout, ntaps, HW, reps=4, which="fwd", stride=1):
    """HIP-event time of every launch of one conv layer shape (Cin -> Cout, ntaps, HW x HW, stride 1) INSIDE the step's own
    launch lists: the list is run in pieces with events around those ops, so each launch reads what the previous kernels
    of the step left behind a 20 GB arena (not a cache-resident toy).  which = "fwd" (forward list), "dgrad" (the data
    gradient: a conv over dz in the backward list) or "wgrad" (filter gradient; run in line on the main stream here).
    Returns (mean ms, launches timed, kernel name chosen by the dispatcher)."""
    from vision_toolbox import _native as N

    p = ts.prog
    ops, n = (p.fwd_ops, p.n_fwd) if which == "fwd" else (p.bwd_ops, p.n_bwd)
    kind = N.OP_CONV_WGRAD if which == "wgrad" else N.OP_CONV_IGEMM
    idxs = []
    for i in range(n):
        op = ops[i]
        if (op.kind & 0xFFFF) != kind:
            continue
        d = N.ConvDesc.from_buffer_copy(bytes(op.i)[: ctypes.sizeof(N.ConvDesc)])
        if (d.Cin, d.Cout, d.ntaps, d.Hi, d.Wi, d.sh) == (Cin, Cout, ntaps, HW, HW, stride):
            idxs.append(i)
    if not idxs:
        return None, 0, ""
    s = int(torch.cuda.current_stream().cuda_stream)
    side = int(ts._side.cuda_stream) if ts._side is not None else 0
    sz = ctypes.sizeof(N.Op)

    def run(lst, lo, hi, use_side):
        if hi > lo:
            sub = (N.Op * (hi - lo)).from_address(ctypes.addressof(lst) + lo * sz)
            N.run_ops(sub, hi - lo, ts.bases, s, side=side if use_side else 0)

    # consecutive matching ops are ONE timed piece: the engine releases the filter gradients of same-shape layers together
    # and the executor hands such a run to vt_conv_wgrad_group (up to 8 layers per launch); per-layer time = piece / layers
    runs = []
    for i in idxs:
        if runs and runs[-1][1] == i:
            runs[-1][1] = i + 1
        else:
            runs.append([i, i + 1])
    pairs, name = [], ""
    for _ in range(reps):
        N.run_ops(ts.zero_ops, 1, ts.bases, s)
        if which != "fwd":
            run(p.fwd_ops, 0, p.n_fwd, True)
        lo = 0
        for i0, i1 in runs:
            run(ops, lo, i0, which == "fwd")
            e0, e1 = N.Event(), N.Event()
            e0.record(s)
            run(ops, i0, i1, False)
            e1.record(s)
            name = N.last_kernel_name()
            pairs.append((e0, e1, i1 - i0))
            lo = i1
        run(ops, lo, n, which == "fwd")
    torch.cuda.synchronize()
    total = sum(a.elapsed_ms(b) for a, b, _ in pairs)
    layers = sum(c for _, _, c in pairs)
    return total / layers, layers, name


def insitu_op_times(ts, which, match, reps=3):
    """HIP-event time of the ops of the forward (`which` = "fwd") or backward list selected by `match(op)`, INSIDE the step's
    own lists (as insitu_layer_times); returns (mean ms, launches timed)"""
    from vision_toolbox import _native as N

    p = ts.prog
    ops, n = (p.fwd_ops, p.n_fwd) if which == "fwd" else (p.bwd_ops, p.n_bwd)
    idxs = [i for i in range(n) if match(ops[i])]
    if not idxs:
        return None, 0
    s = int(torch.cuda.current_stream().cuda_stream)
    side = int(ts._side.cuda_stream) if ts._side is not None else 0
    sz = ctypes.sizeof(N.Op)

    def run(lst, lo, hi, use_side):
        if hi > lo:
            sub = (N.Op * (hi - lo)).from_address(ctypes.addressof(lst) + lo * sz)
            N.run_ops(sub, hi - lo, ts.bases, s, side=side if use_side else 0)

    pairs = []
    for _ in range(reps):
        N.run_ops(ts.zero_ops, 1, ts.bases, s)
        if which != "fwd":
            run(p.fwd_ops, 0, p.n_fwd, True)
        lo = 0
        for i in idxs:
            run(ops, lo, i, which == "fwd")
            e0, e1 = N.Event(), N.Event()
            e0.record(s)
            run(ops, i, i + 1, False)
            e1.record(s)
            pairs.append((e0, e1))
            lo = i + 1
        run(ops, lo, n, which == "fwd")
    torch.cuda.synchronize()
    ms = [a.elapsed_ms(b) for a, b in pairs]
    return sum(ms) / len(ms), len(ms)


def hbm_layers_insitu(ts, batch):
    """the HBM-bound layers of stages 0-1 as the step runs them (HBM fraction on the 1x1 and early-stage convs),
    IN SITU, against 8 TB/s: the short-K 3x3 convs on the persistent resident-filter span kernel (vt_igemm_pspan.hip) and
    the passes of a pointwise 1x1 unit (vt_pointwise.hip; bytes = the operands of the pass, once)"""
    from vision_toolbox import _native as N

    def conv_match(Cin, Cout, ntaps, HW, stride):
        def m(op):
            if (op.kind & 0xFFFF) != N.OP_CONV_IGEMM:
                return False
            d = N.ConvDesc.from_buffer_copy(bytes(op.i)[: ctypes.sizeof(N.ConvDesc)])
            return (d.Cin, d.Cout, d.ntaps, d.Hi, d.Wi, d.sh) == (Cin, Cout, ntaps, HW, HW, stride)
        return m

    def pw_match(kind, K, C0, C1, M):
        def m(op):  # i: K, groups, relu, C0, C1, ldx, ...; f: M
            return (op.kind & 0xFFFF) in kind and (op.i[0], op.i[3], op.i[4]) == (K, C0, C1) and int(op.f[0]) == M
        return m

    out = []
    px = lambda hw: batch * hw * hw
    rows = [
        ("conv3x3 s1 32->32 @112x112 forward (statistics epilogue)", "fwd", conv_match(32, 32, 9, 112, 1), 2.0 * (px(112) * 64 + 32 * 288)),
        ("conv3x3 s2 32->64 @224x224 forward (space-to-depth view)", "fwd", conv_match(32, 64, 9, 224, 2),
         2.0 * (px(224) * 32 + px(112) * 64 + 64 * 288)),
        ("data gradient of conv3x3 s2 32->64: 2x2 taps over dz 64 ch @112x112, depth-to-space store of 32 ch @224x224", "bwd",
         conv_match(64, 128, 4, 112, 1), 2.0 * (px(112) * 64 + px(224) * 32 + 128 * 256)),
        ("conv3x3 s1 64->64 @56x56 forward", "fwd", conv_match(64, 64, 9, 56, 1), 2.0 * (px(56) * 128 + 64 * 576)),
        ("pointwise pair 64->32|32 @112x112, statistics pass (reads x)", "fwd", pw_match((N.OP_PW_STATS,), 64, 32, 32, px(112)), 2.0 * px(112) * 64),
        ("pointwise pair 64->32|32 @112x112, normalise pass (reads x, writes both outputs)", "fwd", pw_match((N.OP_PW_APPLY, N.OP_PW_APPLY_FIN), 64, 32, 32, px(112)),
         2.0 * px(112) * 128),
        ("pointwise pair 64->32|32 @112x112, backward reduction (reads x, dy)", "bwd", pw_match((N.OP_PW_REDUCE,), 64, 32, 32, px(112)),
         2.0 * px(112) * 128),
        ("pointwise pair 64->32|32 @112x112, backward apply (reads x, dy; writes dx; dW in registers)", "bwd",
         pw_match((N.OP_PW_BWD, N.OP_PW_BWD_FIN), 64, 32, 32, px(112)), 2.0 * px(112) * 192),
    ]
    for name, which, match, nbytes in rows:
        ms, n = insitu_op_times(ts, which, match)
        if not ms:
            continue
        out.append({"layer": name + f" B={batch}", "ms": round(ms, 4), "gbs": round(nbytes / ms / 1e6, 1),
                    "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4), "algorithmic_bytes": int(nbytes), "launches_timed": n,
                    "measured": "in situ"})
    return out


def time_train_step(model, batch, image_size, steps, warmup, dev, keep=False):
    """ms per fused train step of `model` at per-GPU batch `batch` on this GPU alone (data_parallel=False: the
    single-GPU program even when this process is a rank of a larger job)"""
    from vision_toolbox import backbones
    from vision_toolbox.trainer import TrainStep

    torch.manual_seed(0)
    ts = TrainStep(getattr(backbones, model)(), 1000, batch, image_size, torch.bfloat16, lr=0.05, momentum=0.9,
                   weight_decay=2e-5, label_smoothing=0.1, device=dev, use_graphs=False, process_group=None,
                   data_parallel=False)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev, generator=g))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev, generator=g))
    for _ in range(warmup):
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    loss = ts.loss()
    if keep:
        return ms, loss, ts
    del ts
    torch.cuda.empty_cache()
    return ms, loss


def layer_rows(ts, batch, shapes):
    """`roofline_layers` rows measured in situ (insitu_layer_times): shapes = (Cin, Cout, taps, H, stride, launches per step,
    which passes).  The data gradient of a layer Cin -> Cout is the stride-1 conv Cout -> Cin over dz."""
    rows = []
    for (ci, co, nt, hw, st, cnt, passes) in shapes:
        ho = hw // st
        fl = 2.0 * batch * ho * ho * co * nt * ci
        row = {"layer": f"conv{'3x3' if nt == 9 else '1x1'} s{st} {ci}->{co} @{hw}x{hw} B={batch} (M={batch * ho * ho} N={co} K={nt * ci})",
               "launches_per_step": cnt, "gflop": round(fl / 1e9, 2)}
        for which in passes:
            a, b = (co, ci) if which == "dgrad" else (ci, co)
            ms_l, n_l, kname = insitu_layer_times(ts, a, b, nt, hw, reps=2, which=which, stride=st)
            if ms_l:
                row[which] = {"ms": round(ms_l, 4), "tflops": round(fl / ms_l / 1e9, 1),
                              "frac": round(fl / ms_l / 1e9 / PEAK_BF16_TFLOPS, 4), "kernel": kname, "launches_timed": n_l}
        rows.append(row)
    return rows


def secondary_configs(dev):
    """BASELINE.json configs[3] and configs[4] in the driver-timed line (secondary numbers)"""
    from vision_toolbox import backbones

    out = []
    # configs[3]: VoVNet-39 forward+backward bf16, batch 256 (the same fused train step); 3 x 15.530 GFLOP per image
    ms, loss, ts = time_train_step("vovnet39", 256, 224, 8, 3, dev, keep=True)
    ts.step()  # (creates the side stream the in-situ pieces run on)
    # VoVNet-39's dominant layer (128 -> 128 3x3 @56x56, five per step, 29.8 % of its MACs) and the first concat 1x1
    vrows = layer_rows(ts, 256, [(128, 128, 9, 56, 1, 5, ("fwd", "dgrad", "wgrad")),
                                 (768, 256, 1, 56, 1, 1, ("fwd", "dgrad", "wgrad"))])
    del ts
    torch.cuda.empty_cache()
    out.append({"config": "BASELINE configs[3]: VoVNet-39 train step (fwd+CE+bwd+SGD) bf16, batch 256 @224",
                "ms": round(ms, 3), "images_per_sec": round(256 / ms * 1e3, 1),
                "tflops": round(3 * 15.530 * 256 / ms, 1), "roofline_frac": round(3 * 15.530 * 256 / ms / PEAK_BF16_TFLOPS, 4),
                "final_loss": round(loss, 4), "roofline_layers": vrows})
    # configs[4]: Darknet-YOLOv5x get_feature_maps() multi-scale forward, batch 64 @640 (module API, eval, no_grad)
    torch.manual_seed(0)
    m = backbones.darknet_yolov5x().to(dev).eval()
    x = torch.rand(64, 3, 640, 640, device=dev)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m.get_feature_maps(x)

    for _ in range(3):
        maps = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        fwd()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    out.append({"config": "BASELINE configs[4]: Darknet-YOLOv5x get_feature_maps() bf16 forward, batch 64 @640",
                "ms": round(ms, 3), "images_per_sec": round(64 / ms * 1e3, 1), "tflops": round(127.599 * 64 / ms, 1),
                "roofline_frac": round(127.599 * 64 / ms / PEAK_BF16_TFLOPS, 4), "maps": [list(t.shape) for t in maps]})
    del m, x, maps
    torch.cuda.empty_cache()
    # the reference recipe's `sync_batchnorm: true` (configs/base.yaml:22) at N = 1: the data-parallel schedule over a
    # one-rank RCCL communicator with the collectives as ops of the launch lists, with and without the 67 + 67
    # statistics exchanges (a child process: it needs a process group of its own)
    sb = syncbn_child()
    if sb:
        out.append(sb)
    return out


def syncbn_child(timeout_s: int = 240):
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_syncbn.py"), "256", "10", "rccl"], env=env,
                           capture_output=True, text=True, timeout=timeout_s)
        line = next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{"))
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001 -- a secondary number: report its absence, do not fail the line
        return {"config": "sync_batchnorm at N=1", "error": repr(e)[:200]}
    plain, sync = min(d["inlist_plain_ms"]), min(d["inlist_sync_bn_ms"])
    return {"config": "CSPDarknet-53 train step, batch 256, `sync_batchnorm: true` (configs/base.yaml:22) at N=1: one-rank RCCL "
                      "communicator, statistics exchanges + bucket all-reduces as launch-list ops (vt_stat_sync / "
                      "vt_allreduce_bucket)",
            "plain_dp_schedule_ms": plain, "sync_bn_step_ms": sync, "sync_bn_overhead": round(sync / plain - 1.0, 4),
            "runs_ms": d}


def pmc_traffic_live(layer: str, kernel_substr: str, timeout_s: int = 150):
    """HBM-side bytes per launch of the dominant kernel, measured NOW: two rocprofv3 child runs
    (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled
    on gfx950) of tools/bench_conv.py on that layer.  None if the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="vt_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("WORLD_SIZE", None)
            subprocess.run([prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--",
                            sys.executable, str(ROOT / "tools" / "bench_conv.py"), "fwd", layer],
                           cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = glob.glob(f"{tmp}/**/*counter_collection.csv", recursive=True)
            vals = []
            for f in files:
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[counter] = sum(vals) / len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    # rocprofv3 reports both in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled
    return {"bytes": out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024,
            "fetch_kb_raw": out["FETCH_SIZE"], "write_kb": out["WRITE_SIZE"]}


def rocprof_launch_avg_live(kernel_substr: str, per_step: int, steps: int = 4, timeout_s: int = 240):
    """The dominant kernel's launch duration as rocprofv3 sees it IN THE STEP, measured now: a child run of this script
    (`--steps-only`: the train steps and nothing else) under `rocprofv3 --kernel-trace`; the dispatches whose name
    contains `kernel_substr` are grouped by (name, LDS bytes, grid) and the group with `per_step` dispatches per step is
    averaged (ties: the largest LDS allocation -- the dominant layer's span is the widest).  The same number can be read
    off profiles/r06_*_bench_kernel_stats.csv; VERDICT r05 #7b: HIP events around the launch see the launch boundary too
    (~4-5 us more), so the line carries both.  None if the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    tmp = tempfile.mkdtemp(prefix="vt_kt_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        warm = 2
        subprocess.run([prof, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
                        str(ROOT / "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--steps-only", "--no-cpu-baseline",
                        "--no-pmc", "--no-secondary"], cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        groups = defaultdict(list)
        for f in glob.glob(f"{tmp}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r.get("Kernel_Name", "").replace(", ", ","):
                    key = (r["Kernel_Name"], int(r.get("LDS_Block_Size", 0) or 0), r.get("Grid_Size", ""))
                    groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        want = per_step * (steps + warm)
        cands = [(k, v) for k, v in groups.items() if len(v) == want]
        if not cands:
            return None
        (name, lds, grid), v = max(cands, key=lambda kv: kv[0][1])
        v = v[per_step * warm:]  # the timed steps
        return {"ms": sum(v) / len(v) / 1e6, "launches": len(v), "kernel_symbol": name[:96], "lds_bytes": lds}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _percentiles(v):
    v = sorted(v)
    pick = lambda q: v[min(len(v) - 1, int(round(q * (len(v) - 1))))]
    return pick(0.1), pick(0.5), pick(0.9)


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: a box that
    SHOWS 128 cores may schedule 16 of them, and a thread per visible core then oversubscribes 8 to 1"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(budget_s: float = 75.0):
    """the oracle (pure torch CPU fp32 restatement of the reference) on the host's cores: CSPDarknet-53
    train step (fwd + CE + bwd + SGD) and Darknet-19 forward at batch 1 (BASELINE configs[0]).  The thread count is SWEPT
    ({8, 16, 32, 64, all physical cores}, bounded by what the box offers): round 4 pinned one thread per visible core and
    measured a baseline four times slower than an 8-core container -- the best count is reported with
    its figure (`threads_best`, `value_best` = `value`) next to the all-cores figure (`value_all_cores`).  10 timed steps
    at batch 32 when the budget allows, else at a smaller batch, and the sample says which."""
    from oracle import filler
    from oracle import torch_ref as R

    cores, cpu_model = host_cpu()
    quota = cgroup_cpu_quota()
    torch.manual_seed(0)
    t_start = time.perf_counter()

    def state(name, ncls):
        sd = {}
        for k, shape in R.classifier_spec(name, ncls).items():
            dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
            sd[k] = filler.fill_tensor("cpu." + k, torch.zeros(shape, dtype=dt))
        return sd

    name, ncls = "cspdarknet53", 1000
    sd = state(name, ncls)
    params = {k: v for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    for v in params.values():
        v.requires_grad_(True)
    mom = {}

    def make_step(bs):
        x, y = filler.images(bs, 224), filler.labels(bs, ncls)

        def step():
            for v in params.values():
                v.grad = None
            loss, _ = R.classifier_loss(name, sd, x, y, 0.1, training=True)
            loss.backward()
            R.sgd_step(params, {k: v.grad for k, v in params.items()}, mom, 0.05, 0.9,
                       lambda k: R.weight_decay_group(k, 2e-5, 0.0, 0.0))
        return step

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n

    sd19 = state("darknet19", ncls)
    x1 = filler.images(1, 224, seed=224)

    def d19():
        with torch.no_grad():
            R.classifier_logits("darknet19", sd19, x1, False)

    cands = sorted({t for t in (8, 16, 32, 64, cores) if t <= cores} | ({int(quota)} if quota and 1 <= quota <= cores else set()))
    sweep = {}
    probe = make_step(8)  # the sweep's own small train step: 1 warm-up + 1 timed per thread count
    for t in cands:
        if time.perf_counter() - t_start > budget_s * 0.45 and sweep:
            break
        torch.set_num_threads(t)
        d19_ms = timed(d19, 2, 5) * 1e3
        tr = timed(probe, 1, 1)
        sweep[t] = {"darknet19_fwd_b1_ms": round(d19_ms, 2), "train_b8_images_per_sec": round(8 / tr, 2)}
    best = max(sweep, key=lambda t: sweep[t]["train_b8_images_per_sec"])
    best19 = min(sweep, key=lambda t: sweep[t]["darknet19_fwd_b1_ms"])
    # the headline sample at the best thread count: 10 timed steps, batch 32 if its projected time fits what is left
    left = budget_s - (time.perf_counter() - t_start)
    rate = sweep[best]["train_b8_images_per_sec"]
    bs = 32
    while bs > 4 and 12 * bs / rate > left * 0.8:
        bs //= 2
    torch.set_num_threads(best)
    sec = timed(make_step(bs), 2, 10)
    value_best = bs / sec
    torch.set_num_threads(best19)
    # (median of 30 single timings: the boxes share their host, a mean over 10 gave 4.9-8.3 ms from run to run)
    for _ in range(3):
        d19()
    singles = []
    for _ in range(30):
        t0 = time.perf_counter()
        d19()
        singles.append((time.perf_counter() - t0) * 1e3)
    singles.sort()
    d19_best = singles[len(singles) // 2]
    all_cores = sweep.get(cores, {}).get("train_b8_images_per_sec")
    torch.set_num_threads(best)
    return {"value": round(value_best, 3), "unit": "images/sec", "cores": best, "kind": "port",
            "threads_best": best, "value_best": round(value_best, 3), "value_all_cores": all_cores,
            "physical_cores": cores, "cgroup_cpu_quota": quota, "thread_sweep": {str(k): v for k, v in sweep.items()},
            "cpu_model": cpu_model, "logical_cpus": os.cpu_count(),
            "darknet19_fwd_b1_ms": round(d19_best, 2), "darknet19_threads": best19,
            "darknet19_fwd_b1_ms_p10_p90": [round(singles[3], 2), round(singles[26], 2)],
            "sample": f"oracle/torch_ref.py CSPDarknet-53 fp32 train step (fwd+CE+bwd+SGD), batch {bs} @224, 10 timed steps "
                      f"after 2 warm-up at {best} threads (the best of the sweep {list(sweep)}; sweep entries: batch-8 step, "
                      f"1 warm-up + 1 timed; value_all_cores = the sweep entry at {cores} threads); Darknet-19 forward batch 1 "
                      f"@224 (BASELINE configs[0]) median of 30 timed after 3 warm-up at {best19} threads; torch {torch.__version__} CPU; "
                      f"{time.perf_counter() - t_start:.0f}s of CPU work in all"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # >= 20 warm-up, >= 100 timed
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch; default 256 for every N (BASELINE configs[1] at N = 1; N > 1 is its WEAK scaling: the "
                         "per-GPU work of the N = 1 line, global batch 256*N)")
    ap.add_argument("--config3-batch", type=int, default=None,
                    help="N > 1: per-GPU batch of the additional BASELINE configs[2] measurement carried by the same line "
                         "(`config3`; default 128 = global batch 1024 on 8 GPUs when --batch is left at its default, else off; 0: off)")
    ap.add_argument("--no-secondary", action="store_true", help="skip configs[3] / configs[4] and the batch-128 step")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "sharded"],
                    help="gradient exchange: f32 all-reduce per bucket (default) or reduce-scatter -> sharded SGD -> "
                         "bf16 all-gather of the weights (SURVEY 8e)")
    ap.add_argument("--collectives", default="auto", choices=["auto", "torch", "rccl"],
                    help="who issues the collectives for N > 1: torch.distributed between list segments, or the library's own "
                         "RCCL communicator with the collectives as ops of the launch lists (auto: rccl on the nccl backend "
                         "with the all-reduce exchange)")
    ap.add_argument("--model", default="cspdarknet53")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--graphs", action="store_true",
                    help="replay captured hipGraphs (default: native executor with the filter gradients on a "
                         "side stream, which measured faster: hipGraph serialises the forked branches)")
    ap.add_argument("--no-graphs", action="store_true", help="(default) kept for compatibility")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=16.0)
    ap.add_argument("--sync-bn", action="store_true",
                    help="SyncBatchNorm as in the reference recipe (configs/base.yaml:22); off for the headline "
                         "metric: 2 small sequential collectives per unit")
    ap.add_argument("--main-priority", type=int, default=None,
                    help="run the step on a torch stream of this priority (-1 = high): the filter-gradient side "
                         "stream then only fills what the critical path leaves.  Default: -1 on one GPU (measured "
                         "21.86 -> 21.71 and 22.14 -> 22.00 ms on two boxes), 0 with --gpus N > 1, where the RCCL "
                         "kernels of the gradient exchange run on normal-priority streams and must not be starved "
                         "by the step (not measurable on a one-GPU box)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 child runs that measure `traffic`")
    ap.add_argument("--deterministic", action="store_true",
                    help="bit-identical parameter updates (two-stage filter gradients, fixed-point bias sums): ~3 %% slower")
    ap.add_argument("--steps-only", action="store_true",
                    help="(profiling) run warm-up + timed steps and print their time; no roofline / baseline launches, so a "
                         "rocprofv3 pass over this command counts the step's kernels and nothing else")
    ap.add_argument("--plan-only", action="store_true",
                    help="(tests) build the launch lists and the bucket plan on the CPU, run the first collectives "
                         "over the given backend and exit: exercises the N>1 launch path without a GPU")
    args = ap.parse_args()
    if args.config3_batch is None:
        args.config3_batch = 128 if args.batch is None else 0
    if args.batch is None:
        # Round 5: 256 per GPU for EVERY N.  The driver derives the scaling efficiency from the per-N `value`s of this command;
        # until round 4 the N > 1 lines ran BASELINE configs[2] (128 per GPU) against an N = 1 line at 256, so that quotient
        # mixed the scaling with the batch-size effect on one GPU (12.1 ms at 128 = 0.83 of the rate at 256).  Now the series
        # is weak scaling in the sense of the contract (per-GPU work fixed), and configs[2] is measured IN THE SAME RUN as
        # `config3`, with its own single-GPU denominator.
        args.batch = 256

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # before anything touches the GPU

    from vision_toolbox import _native as N
    from vision_toolbox import backbones
    from vision_toolbox.distributed import init_from_env
    from vision_toolbox.trainer import TrainStep

    # test hooks (1-GPU boxes): VT_DIST_BACKEND=gloo VT_FORCE_DEVICE=0 let two ranks share one GPU so the
    # N>1 control flow of this script can be exercised without RCCL; never set them for a measurement
    backend = os.environ.get("VT_DIST_BACKEND", "nccl")
    if "VT_FORCE_DEVICE" in os.environ:
        os.environ["LOCAL_RANK"] = os.environ["VT_FORCE_DEVICE"]
    rank, local, world = init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.plan_only:
        torch.manual_seed(0)
        ts = TrainStep(getattr(backbones, args.model)(), 1000, args.batch, args.image_size, torch.bfloat16,
                       device="cpu", bucket_mb=args.bucket_mb, plan_only=True, sync_bn=args.sync_bn, exchange=args.exchange)
        ts.store.pflat.add_(float(rank))  # ranks start different; the broadcast must make them equal
        ts.broadcast_parameters(0)
        ts.gflat.fill_(float(rank + 1))
        ts.bucketer.reduce_all()
        ts.bucketer.finish()
        want = world * (world + 1) / 2
        if args.exchange == "sharded":  # reduce-scatter: a rank holds the sums of its own slices
            ok = all(bool((ts.gflat[a:b] == want).all()) for a, b in ts.bucketer.shards)
        else:
            ok = bool((ts.gflat == want).all())
        chk = torch.tensor([float(ts.store.pflat.double().sum())], dtype=torch.float64)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        # the N > 1 line is self-contained: rank 0 alone builds (and, on a GPU, times) the single-GPU program at the same
        # per-GPU batch while the other ranks wait at a barrier; here only the control flow and the plan are exercised
        n1_segments = n1_launches = None
        if rank == 0:
            torch.manual_seed(0)
            solo = TrainStep(getattr(backbones, args.model)(), 1000, args.batch, args.image_size, torch.bfloat16,
                             device="cpu", plan_only=True, data_parallel=False)
            assert solo.world == 1 and solo.bucketer is None
            n1_segments, n1_launches = len(solo.bwd_cuts), solo.prog.n_fwd + solo.prog.n_bwd
            assert n1_launches == ts.prog.n_fwd + ts.prog.n_bwd, "a rank must run the single-GPU launch lists"
        dist.barrier()
        if rank == 0:
            print(json.dumps({"plan_only": True, "n_gpus": world, "backend": backend, "buckets": len(ts.bucketer.buckets),
                              "n1_same_per_gpu_batch_ms": None, "n1_same_per_gpu_batch_images_per_sec": None,
                              "weak_scaling_efficiency": None, "exchange_exposed_ms": None,
                              "n1_plan": {"bwd_segments": n1_segments, "launches": n1_launches},
                              "bwd_segments": len(ts.bwd_cuts), "allreduce_ok": ok, "gradient_exchange": ts.exchange,
                              "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                              "head_bucket_bytes": 4 * (ts.bucketer.buckets[-1][1] - ts.bucketer.buckets[-1][0]),
                              "params_equal": bool(lo.item() == hi.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    bb = getattr(backbones, args.model)()
    coll = args.collectives
    if coll == "auto":
        # (VT_BENCH_AUTO_RCCL=1: test hook -- take the rccl branch under the gloo test backend too: two ranks on one GPU are
        #  refused by RCCL, which exercises the agreed fall-back below)
        nccl_like = backend == "nccl" or os.environ.get("VT_BENCH_AUTO_RCCL", "0") == "1"
        coll = "rccl" if (world > 1 and nccl_like and args.exchange == "allreduce" and not args.graphs) else "torch"
    if coll == "rccl" and args.collectives == "auto":
        # the library's communicator is collective to create: agree on the outcome, and fall back to torch.distributed on
        # EVERY rank if it could not be made on any of them (the line reports which form ran: config.collectives)
        from vision_toolbox.distributed import ensure_library_comm, self_test_library_comm

        ok = 1
        try:
            # (agreed on every rank by itself: a rank that cannot bind RCCL raises everywhere, nobody blocks)
            ensure_library_comm(None, dev, stat_comm=args.sync_bn)
            # ... and one real collective through it, under a host-side watchdog, before the step depends on it: this path
            # has never run with more than one rank on this pool (one-GPU boxes; RCCL refuses two ranks on one device)
            if not self_test_library_comm(dev):
                raise RuntimeError("the library communicator's first all-reduce did not complete (or summed wrongly)")
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench rank {rank}] library RCCL communicator unavailable ({e!r}); falling back to torch.distributed",
                  file=sys.stderr, flush=True)
        flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            coll = "torch"
            if ok:  # (this rank is healthy, another is not: leave the communicator alone if its self-test hung elsewhere)
                try:
                    N.lib().vt_comm_destroy()
                except Exception:  # noqa: BLE001
                    pass
    ts = TrainStep(bb, 1000, args.batch, args.image_size, torch.bfloat16, lr=0.05, momentum=0.9, weight_decay=2e-5,
                   label_smoothing=0.1, device=dev, bucket_mb=args.bucket_mb, use_graphs=args.graphs,
                   sync_bn=args.sync_bn, deterministic=True if args.deterministic else None, exchange=args.exchange,
                   collectives=coll)
    ts_collectives = ts.collectives
    ts.broadcast_parameters(0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev, generator=g))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev, generator=g))

    def barrier():
        if world > 1:
            dist.barrier()

    if args.main_priority is None:
        args.main_priority = -1 if world == 1 else 0
    if args.main_priority != 0:
        torch.cuda.set_stream(torch.cuda.Stream(dev, priority=args.main_priority))
    launches0 = N.launch_count()
    for _ in range(args.warmup):
        ts.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s_handle = int(torch.cuda.current_stream().cuda_stream)
    marks = [N.Event() for _ in range(args.steps + 1)]
    marks[0].record(s_handle)
    for i in range(args.steps):
        ts.step()
        marks[i + 1].record(s_handle)  # no host sync: read back after the timed region
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = ts.loss()
    assert N.launch_count() > launches0 and loss == loss, "HIP path did not run / loss is NaN"

    # ---- N > 1: the line carries its own denominators (no RCCL N > 1 run exists on a one-GPU box: DESIGN 6) ---------
    exposed_ms = n1_ms = None
    if world > 1 and not args.steps_only:
        # (a) what the gradient exchange leaves exposed: the same data-parallel schedule (cut lists, stream ordering)
        #     with the collectives not issued, every rank, max over ranks
        with ts.exchange_skipped():  # (restores parameters / momentum / BatchNorm state afterwards: ADVICE r04)
            for _ in range(3):
                ts.step()
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ts.step()
            torch.cuda.synchronize()
            el_nx = time.perf_counter() - t1
        t = torch.tensor([el_nx], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = (elapsed - float(t.item())) / args.steps * 1e3
        # (b) the weak-scaling denominator: rank 0 alone times the single-GPU program at the same per-GPU batch (same
        #     stream priority as the run above) while the other ranks wait at the barrier
        if rank == 0:
            n1_ms, _ = time_train_step(args.model, args.batch, args.image_size, args.steps, args.warmup, dev)
        barrier()

    # ---- N > 1: BASELINE configs[2] itself (128 images per GPU: global batch 1024 on 8 GPUs) in the same run ---------------
    config3 = None
    if world > 1 and not args.steps_only and args.config3_batch > 0 and args.config3_batch != args.batch:
        b3 = args.config3_batch
        k3, w3 = min(args.steps, 50), min(args.warmup, 10)
        torch.manual_seed(0)
        ts3 = TrainStep(getattr(backbones, args.model)(), 1000, b3, args.image_size, torch.bfloat16, lr=0.05, momentum=0.9,
                        weight_decay=2e-5, label_smoothing=0.1, device=dev, bucket_mb=args.bucket_mb, use_graphs=args.graphs,
                        sync_bn=args.sync_bn, deterministic=True if args.deterministic else None, exchange=args.exchange,
                        collectives=coll)
        ts3.broadcast_parameters(0)
        ts3.images.copy_(torch.rand(ts3.images.shape, device=dev, generator=g))
        ts3.labels.copy_(torch.randint(0, 1000, ts3.labels.shape, device=dev, generator=g))
        for _ in range(w3):
            ts3.step()
        torch.cuda.synchronize()
        barrier()
        t3 = time.perf_counter()
        for _ in range(k3):
            ts3.step()
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t3
        t = torch.tensor([el3], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el3 = float(t.item())
        loss3 = ts3.loss()
        del ts3
        torch.cuda.empty_cache()
        n1_3 = None
        if rank == 0:
            n1_3, _ = time_train_step(args.model, b3, args.image_size, k3, w3, dev)
        barrier()
        if rank == 0:
            v3 = b3 * world * k3 / el3
            config3 = {"config": f"BASELINE configs[2]: {args.model} data-parallel train step, {b3} images per GPU, global batch "
                                 f"{b3 * world}, same run, same ranks and communicator",
                       "per_gpu_batch": b3, "global_batch": b3 * world, "steps": k3, "warmup": w3,
                       "ms_per_step": round(el3 / k3 * 1e3, 3), "images_per_sec": round(v3, 1), "final_loss": round(loss3, 4),
                       "n1_same_per_gpu_batch_ms": round(n1_3, 3),
                       "n1_same_per_gpu_batch_images_per_sec": round(b3 / n1_3 * 1e3, 1),
                       "weak_scaling_efficiency": round(v3 / (world * b3 / n1_3 * 1e3), 4)}

    if args.steps_only:
        if rank == 0:
            print(json.dumps({"steps_only": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "final_loss": round(loss, 4)}), flush=True)
        if world > 1:
            torch.cuda.synchronize()
            N.lib().vt_comm_destroy()  # (no-op without the library communicator)
            dist.destroy_process_group()
        return
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        # Dominant kernel: the two-group + loader-wave span kernel (vt_igemm_span6.hip) on the stride-1 3x3 convs and
        # their data gradients.  Every roofline entry of the three dominant 3x3 shapes (128 -> 128 @28x28 x8,
        # 256 -> 256 @14x14 x8, 512 -> 512 @7x7 x4) is measured IN SITU: events around those launches inside the step's own
        # forward / backward lists, for the forward conv, the data gradient and the filter gradient.
        scale = args.image_size / 224.0
        shapes = [(128, 128, 9, int(round(28 * scale)), 8), (256, 256, 9, int(round(14 * scale)), 8),
                  (512, 512, 9, int(round(7 * scale)), 4)] if args.model in ("cspdarknet53", "darknet53") else []
        layers, tot_fl, tot_ms, insitu = [], 0.0, 0.0, None
        for (ci, co, nt, hw, cnt) in shapes:
            fl = 2.0 * args.batch * hw * hw * co * nt * ci
            row = {"layer": f"conv3x3 s1 {ci}->{co} @{hw}x{hw} B={args.batch} (M={args.batch * hw * hw} N={co} K={nt * ci})",
                   "launches_per_step": cnt, "gflop": round(fl / 1e9, 2)}
            for which in ("fwd", "dgrad", "wgrad"):
                ms_l, n_l, kname = insitu_layer_times(ts, ci, co, nt, hw, reps=3, which=which)
                if not ms_l:
                    continue
                row[which] = {"ms": round(ms_l, 4), "tflops": round(fl / ms_l / 1e9, 1),
                              "frac": round(fl / ms_l / 1e9 / PEAK_BF16_TFLOPS, 4), "kernel": kname, "launches_timed": n_l}
                tot_fl += cnt * fl
                tot_ms += cnt * ms_l
                if which == "fwd" and insitu is None:
                    insitu = {"ms": ms_l, "n": n_l, "kernel": kname, "tflops": fl / ms_l / 1e9, "flops": fl, "shape": row["layer"]}
            layers.append(row)
        if args.model == "cspdarknet53" and args.image_size == 224:
            # the stride-2 3x3 convs that open the stages (24.8 % of the MACs): forward and filter gradient in situ
            layers += layer_rows(ts, args.batch, [(32, 64, 9, 224, 2, 1, ("fwd", "wgrad")), (64, 128, 9, 112, 2, 1, ("fwd", "wgrad")),
                                                  (128, 256, 9, 56, 2, 1, ("fwd", "wgrad")), (256, 512, 9, 28, 2, 1, ("fwd", "wgrad")),
                                                  (512, 1024, 9, 14, 2, 1, ("fwd", "wgrad"))])
        standalone = conv_roofline(args.batch, 128, 28, N.VT_BF16)
        standalone_name = N.last_kernel_name()
        dom = insitu or {"ms": standalone["ms"], "n": 30, "kernel": standalone_name, "tflops": standalone["tflops"],
                         "flops": standalone["flops"], "shape": standalone["shape"]}
        traffic = None
        if world == 1 and not args.no_pmc:
            # (the rocprof rows are matched by the name the dispatcher reported for this layer, e.g. "span6_kernel")
            traffic = pmc_traffic_live("128,128,3,1,28", (standalone_name or dom["kernel"]).split("<")[0])
        rocprof_avg = None
        if world == 1 and not args.no_pmc and insitu is not None and args.model == "cspdarknet53" and args.batch == 256:
            # (MODE 1 = the statistics epilogue = the forward launches; 8 DarknetBlock.conv2 units of stage 2 per step)
            rocprof_avg = rocprof_launch_avg_live("span6_kernel<1,", 8)
        # HBM-bound layers of stages 0-1 as the step runs them, in situ (HBM fraction on the early-stage convs)
        hbm_layers = hbm_layers_insitu(ts, args.batch) if args.model == "cspdarknet53" and args.image_size == 224 else []
        cfg = ("BASELINE configs[1]" if (world == 1 and args.batch == 256) else
               (f"weak scaling of BASELINE configs[1]: data parallel over RCCL, {args.batch} images per GPU, global batch "
                f"{args.batch * world}; BASELINE configs[2] ({args.config3_batch} per GPU) in `config3`" if config3 else
                f"data parallel over RCCL, {args.batch} images per GPU, global batch {args.batch * world}")
               if world > 1 else f"single GPU, batch {args.batch}")
        out = {
            "metric": f"images/sec (node) {LABELS.get(args.model, args.model)} bf16 train step @{args.image_size}px",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model} train step (fwd+CE+bwd+allreduce+SGD), batch {args.batch}/GPU, "
                                   f"3x{args.image_size}x{args.image_size}, 1000 classes, {cfg}",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "rccl_ranks": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
                       "backend": backend if world > 1 else None, "local_device": f"cuda:{local}",
                       "gradient_exchange": args.exchange if world > 1 else None,
                       "collectives": (ts_collectives if world > 1 else None),
                       "hip_graphs": bool(args.graphs), "main_stream_priority": args.main_priority,
                       "sync_bn": bool(args.sync_bn), "deterministic": bool(ts.deterministic),
                       "final_loss": round(loss, 4)},
            "roofline": {"bound": "mfma", "achieved": round(dom["tflops"], 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_BF16_TFLOPS, 4),
                         # MAC-weighted over the three dominant 3x3 shapes x {forward, data gradient, filter gradient}:
                         # total FLOPs of those launches / their total in-situ time / peak -- not the best case
                         "frac_fwd_dgrad_wgrad": round(tot_fl / tot_ms / 1e9 / PEAK_BF16_TFLOPS, 4) if tot_ms else None,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH x2 on gfx950) "
                                            "run by this bench on tools/bench_conv.py: STANDALONE launches of this layer "
                                            "back to back (x and the filter re-read from the memory-side cache where the step's "
                                            "own launch finds what the previous kernels left), while launch_ms / frac are in situ"
                                            if traffic else None),
                         "algorithmic_bytes": 2.0 * (args.batch * 28 * 28 * 256 + 128 * 1152),
                         "kernel": dom["kernel"], "launch_ms": round(dom["ms"], 4), "launches_timed": dom["n"],
                         # the same launches as rocprofv3's kernel trace of the step sees them (kernel start to end,
                         # without the launch boundary the in-situ HIP events include) and the fraction that follows
                         "launch_ms_rocprof": round(rocprof_avg["ms"], 4) if rocprof_avg else None,
                         "frac_rocprof": round(dom["flops"] / (rocprof_avg["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
                         if rocprof_avg else None,
                         "rocprof_source": ({k: rocprof_avg[k] for k in ("launches", "kernel_symbol", "lds_bytes")}
                                            if rocprof_avg else None),
                         "measured": "in situ (events around the layer's launches inside the step's forward list)"
                         if insitu else "standalone", "layer": dom["shape"]},
            "roofline_layers": layers,
            "ms_per_step_p10_p50_p90": [round(float(v), 3) for v in
                                        _percentiles([marks[i].elapsed_ms(marks[i + 1]) for i in range(args.steps)])],
            "roofline_hbm_layers": hbm_layers,
            "train_step_tflops": round(28.0e9 * (args.batch / 1.0) * world / (ms * 1e-3) / 1e12, 1)
            if args.model == "cspdarknet53" and args.image_size == 224 else None,
        }
        if world > 1:
            n1_ips = args.batch / n1_ms * 1e3
            out["n1_same_per_gpu_batch_ms"] = round(n1_ms, 3)
            out["n1_same_per_gpu_batch_images_per_sec"] = round(n1_ips, 1)
            out["weak_scaling_efficiency"] = round(value / (world * n1_ips), 4)
            out["exchange_exposed_ms"] = round(exposed_ms, 3)
            out["exchange_exposed_note"] = ("ms_per_step minus the same data-parallel schedule timed with the collectives "
                                            "not issued (max over ranks); the N = 1 figures are rank 0 alone, same per-GPU batch")
            if config3:
                out["config3"] = config3
        if world == 1 and not args.no_secondary:
            del ts
            torch.cuda.empty_cache()
            if args.batch != 128:
                # the weak-scaling denominator of configs[2]: the same step at 128 images on this one GPU (and the
                # place where host-issue limits would show: half the GPU time per step, the same 600 launches)
                ms128, _ = time_train_step(args.model, 128, args.image_size, args.steps, args.warmup, dev)
                out["n1_same_per_gpu_batch_ms"] = round(ms128, 3)
                out["n1_same_per_gpu_batch_images_per_sec"] = round(128 / ms128 * 1e3, 1)
            out["secondary"] = secondary_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        torch.cuda.synchronize()
        N.lib().vt_comm_destroy()  # (no-op without the library communicator)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
