"""Probe (GPU box): what a collective library's channel kernels cost the train step on ONE GPU.

    python tools/rccl_hog.py [batch=128] [steps=20] [mode=all|nohog|hog]

Round 6 form.  What the probe measures is exactly what it sets:

* the step's MAIN stream is an explicit torch stream of priority 0 (what `bench.py --gpus N` sets for N > 1) and, as a
  second series, of priority -1 (what it sets at N = 1); the library's filter-gradient stream is the one TrainStep makes;
* the hog sits WHERE THE COLLECTIVE WILL SIT: the step runs its data-parallel schedule over a one-rank gloo group
  (VT_DP_WORLD1=1, collectives="torch": cut launch lists, `bucketer.reduce_bucket` issued with the filter-gradient stream
  current behind the op that completes the bucket) and `reduce_bucket` is replaced by a launch of the hog on that stream
  -- between the filter-gradient launches, ordered like VT_OP_ALLREDUCE / the torch all-reduce would be.  Duration per
  bucket = bucket bytes / 104 MiB x 1.25 ms (ring all-reduce of the whole gradient over xGMI) and 2x that;
* the hog's workgroups are sized like RCCL's channel kernels: 256 or 512 threads, 64 KiB of LDS, W in {8, 16, 32};
* as a control the same hog on a THIRD stream (the round-5 form) for the whole window.

mode=nohog / mode=hog run ten steps of one configuration only (no sweep): for `rocprofv3 --kernel-trace`, whose per-launch
durations of span6_kernel / wgrad6_kernel / pspan_kernel with and without a resident hog answer "does a CU-owning launch
fall into two rounds" as a histogram (tools/hog_hist.py) instead of through the step time.

The hog kernel lives in tools/diag/vt_diag_hog.hip (diagnostics only; built here on demand), not in libvt_amd.so."""
import ctypes
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch
import torch.distributed as dist

from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


def hog_lib():
    so, src = ROOT / "tools" / "diag" / "libvt_diag_hog.so", ROOT / "tools" / "diag" / "vt_diag_hog.hip"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", str(src), "-o", str(so)],
                       check=True)
    lib = ctypes.CDLL(str(so))
    lib.vt_diag_hog.restype = ctypes.c_int
    lib.vt_diag_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    mode = sys.argv[3] if len(sys.argv) > 3 else "all"
    dev = torch.device("cuda")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29647")
    os.environ["VT_DP_WORLD1"] = "1"
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.manual_seed(0)
    ts = TrainStep(backbones.cspdarknet53(), 1000, B, 224, torch.bfloat16, lr=0.05, momentum=0.9, weight_decay=2e-5,
                   label_smoothing=0.1, device=dev, bucket_mb=16.0, collectives="torch")
    assert ts.dp and ts.bucketer is not None and not ts.use_graphs
    ts.images.copy_(torch.rand(ts.images.shape, device=dev))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev))
    H = hog_lib()
    buckets = list(ts.bucketer.buckets)
    total = float(sum(b1 - b0 for b0, b1 in buckets))
    cfg = {"wgs": 0, "threads": 256, "lds": 64 * 1024, "ms_total": 1.25, "third": None}
    real_reduce, real_finish = ts.bucketer.reduce_bucket, ts.bucketer.finish

    def fake_reduce(bi):
        # (called with the filter-gradient stream current, behind a wait for the main stream's position: trainer._run_list)
        if cfg["wgs"] and cfg["third"] is None:
            b0, b1 = buckets[bi]
            us = (b1 - b0) / total * cfg["ms_total"] * 1e3
            rc = H.vt_diag_hog(cfg["wgs"], cfg["threads"], cfg["lds"], us, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

    ts.bucketer.reduce_bucket = fake_reduce
    ts.bucketer.finish = lambda: None
    third = torch.cuda.Stream(device=dev)

    def run(prio, nsteps=steps):
        main_stream = torch.cuda.Stream(device=dev, priority=prio)
        with torch.cuda.stream(main_stream):
            for _ in range(5):
                ts.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                if cfg["third"] is not None and cfg["wgs"]:
                    # round-5 control: a hog on a third stream, started with the step, delayed into backward on one workgroup
                    third.wait_stream(main_stream)
                    H.vt_diag_hog(1, 64, 0, cfg["third"], third.cuda_stream)
                    H.vt_diag_hog(cfg["wgs"], cfg["threads"], cfg["lds"], cfg["ms_total"] * 1e3, third.cuda_stream)
                ts.step()
                if cfg["third"] is not None and cfg["wgs"]:
                    main_stream.wait_stream(third)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / nsteps * 1e3

    if mode in ("nohog", "hog"):
        cfg.update(wgs=32 if mode == "hog" else 0, threads=512, ms_total=2.5)
        ms = run(0, 10)
        print(json.dumps({"mode": mode, "batch": B, "step_ms": round(ms, 3), "cfg": {k: v for k, v in cfg.items()}}), flush=True)
        return

    out = {"batch": B, "steps": steps, "buckets": len(buckets), "segments": len(ts.bwd_cuts), "series": []}
    for prio in (0, -1):
        cfg.update(wgs=0, third=None)
        base = run(prio)
        rows = {"main_stream_priority": prio, "step_ms_no_hog": round(base, 3), "on_filter_gradient_stream": [], "on_third_stream": []}
        for threads in (256, 512):
            for w in (8, 16, 32):
                for ms_total in (1.25, 2.5):
                    cfg.update(wgs=w, threads=threads, ms_total=ms_total, third=None)
                    ms = run(prio)
                    rows["on_filter_gradient_stream"].append({"wgs": w, "threads": threads, "lds_kib": 64, "hog_ms_per_step": ms_total,
                                                              "step_ms": round(ms, 3), "slowdown": round(ms / base - 1, 4)})
        for w in (16, 32):
            cfg.update(wgs=w, threads=512, ms_total=1.5, third=base * 1e3 * 0.55)
            ms = run(prio)
            rows["on_third_stream"].append({"wgs": w, "threads": 512, "lds_kib": 64, "hog_ms_per_step": 1.5,
                                            "step_ms": round(ms, 3), "slowdown": round(ms / base - 1, 4)})
        cfg.update(wgs=0, third=None)
        rows["step_ms_no_hog_again"] = round(run(prio), 3)
        out["series"].append(rows)
        print(json.dumps(rows), flush=True)
    print(json.dumps(out), flush=True)
    ts.bucketer.reduce_bucket, ts.bucketer.finish = real_reduce, real_finish
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
