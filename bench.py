#!/usr/bin/env python
"""Headline benchmark: images/sec of one CSPDarknet-53 bf16 train step @224px.

    python bench.py --gpus N --steps K --warmup W

A step = forward + label-smoothing CE + backward + gradient all-reduce (N>1) + SGD(momentum)
over one synthetic batch resident in HBM (BASELINE.json configs[1]: batch 256 per GPU,
3x224x224, uniform [0,1) images, random-init weights; weak scaling for N>1).
Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel
(measured live with HIP events on the launch stream) and the CPU baseline (the oracle,
timed on the host cores on a bounded sample).
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
    stats = torch.zeros(N.VT_STAT_REPLICAS, 2, Cout, device=dev)
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


def _percentiles(v):
    v = sorted(v)
    pick = lambda q: v[min(len(v) - 1, int(round(q * (len(v) - 1))))]
    return pick(0.1), pick(0.5), pick(0.9)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC pass
    (profiles/r01_dominant_kernel_pmc.json: FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate runs)."""
    f = ROOT / "profiles" / "r01_dominant_kernel_pmc.json"
    try:
        return float(json.loads(f.read_text())["traffic_bytes"])
    except Exception:
        return None


def cpu_baseline(seconds: float = 12.0):
    """oracle (pure torch CPU fp32 restatement of the reference) CSPDarknet-53 train step."""
    from oracle import filler
    from oracle import torch_ref as R

    torch.manual_seed(0)
    name, ncls, bs = "cspdarknet53", 1000, 8
    sd = {}
    for k, shape in R.classifier_spec(name, ncls).items():
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = filler.fill_tensor("cpu." + k, torch.zeros(shape, dtype=dt))
    params = {k: v for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    for v in params.values():
        v.requires_grad_(True)
    x, y = filler.images(bs, 224), filler.labels(bs, ncls)
    mom = {}

    def step():
        for v in params.values():
            v.grad = None
        loss, _ = R.classifier_loss(name, sd, x, y, 0.1, training=True)
        loss.backward()
        R.sgd_step(params, {k: v.grad for k, v in params.items()}, mom, 0.05, 0.9,
                   lambda k: R.weight_decay_group(k, 2e-5, 0.0, 0.0))

    step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 2:
            break
    return {"value": round(bs * n / el, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/torch_ref.py CSPDarknet-53 fp32 train step (fwd+CE+bwd+SGD), batch {bs} @224, "
                      f"{n} steps in {el:.1f}s after 1 warm-up, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
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
    ap.add_argument("--main-priority", type=int, default=0,
                    help="run the step on a torch stream of this priority (-1 = high): the filter-gradient side "
                         "stream then only fills what the critical path leaves")
    args = ap.parse_args()

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    bb = getattr(backbones, args.model)()
    ts = TrainStep(bb, 1000, args.batch, args.image_size, torch.bfloat16, lr=0.05, momentum=0.9, weight_decay=2e-5,
                   label_smoothing=0.1, device=dev, bucket_mb=args.bucket_mb, use_graphs=args.graphs,
                   sync_bn=args.sync_bn)
    ts.broadcast_parameters(0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev, generator=g))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev, generator=g))

    def barrier():
        if world > 1:
            dist.barrier()

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

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        # dominant kernel: span_kernel<bf16,224,128> (vt_igemm_span.hip; 224-row tiles at this pixel count) on the stride-1 3x3 convs and
        # their data gradients; roofline on the layer shape with the largest share of the step
        layers = [conv_roofline(args.batch, 128, 28, N.VT_BF16), conv_roofline(args.batch, 256, 14, N.VT_BF16),
                  conv_roofline(args.batch, 512, 7, N.VT_BF16)]
        dom = layers[0]
        # HBM-bound layers of stages 0-2 (HBM fraction on the 1x1 and early-stage convs)
        hbm_layers = [conv_roofline(args.batch, 64, 112, N.VT_BF16, k=1), conv_roofline(args.batch, 128, 56, N.VT_BF16, k=1),
                      conv_roofline(args.batch, 8, 224, N.VT_BF16, k=3, Cout=32)]
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
                                   f"3x{args.image_size}x{args.image_size}, 1000 classes, BASELINE configs[1]",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "hip_graphs": bool(args.graphs), "sync_bn": bool(args.sync_bn), "final_loss": round(loss, 4)},
            "roofline": {"bound": "mfma", "achieved": round(dom["tflops"], 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic(),
                         "kernel": "span_kernel<bf16,224,128,2,2>", "launch_ms": round(dom["ms"], 4),
                         "layer": dom["shape"]},
            "roofline_layers": [{"layer": l["shape"], "ms": round(l["ms"], 4), "tflops": round(l["tflops"], 1),
                                 "frac": round(l["tflops"] / PEAK_BF16_TFLOPS, 4)} for l in layers],
            "ms_per_step_p10_p50_p90": [round(float(v), 3) for v in
                                        _percentiles([marks[i].elapsed_ms(marks[i + 1]) for i in range(args.steps)])],
            "roofline_hbm_layers": [{"layer": l["shape"], "ms": round(l["ms"], 4), "gbs": round(l["gbs"], 1),
                                     "frac": round(l["gbs"] / PEAK_HBM_GBS, 4), "algorithmic_bytes": int(l["bytes"])}
                                    for l in hbm_layers],
            "train_step_tflops": round(28.0e9 * (args.batch / 1.0) * world / (ms * 1e-3) / 1e12, 1)
            if args.model == "cspdarknet53" and args.image_size == 224 else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
