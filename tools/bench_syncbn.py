"""SyncBatchNorm (configs/base.yaml:22) cost on ONE GPU: the data-parallel schedule over a one-rank RCCL group, plain
BatchNorm against sync_bn=True (67 + 67 statistics exchanges per step), the collectives through torch.distributed
between list segments or as ops of the lists (collectives="rccl").

    python tools/bench_syncbn.py [batch] [steps] [both|rccl]

Prints one JSON line (bench.py reads it for `secondary`: `rccl` = the in-list collectives only)."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
import torch.distributed as dist


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    os.environ["VT_DP_WORLD1"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from vision_toolbox import backbones
    from vision_toolbox.trainer import TrainStep

    out = {}
    which = sys.argv[3] if len(sys.argv) > 3 else "both"
    cases = [(False, "rccl"), (True, "rccl")] + ([(False, "torch"), (True, "torch")] if which == "both" else [])
    for sync, coll in cases + cases:
        torch.manual_seed(0)
        ts = TrainStep(backbones.cspdarknet53(), 1000, batch, 224, torch.bfloat16, lr=0.05, device="cuda", use_graphs=False,
                       sync_bn=sync, collectives=coll)
        ts.images.uniform_()
        ts.labels.random_(0, 1000)
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out.setdefault((sync, coll), []).append(round(ms, 3))
        del ts
        torch.cuda.empty_cache()
    res = {"batch": batch, "inlist_plain_ms": out[(False, "rccl")], "inlist_sync_bn_ms": out[(True, "rccl")]}
    if which == "both":
        res.update({"torch_plain_dp_schedule_ms": out[(False, "torch")], "torch_sync_bn_ms": out[(True, "torch")]})
    print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
