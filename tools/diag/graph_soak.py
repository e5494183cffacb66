"""Soak (GPU box): the captured-graph form of a train step, N create / replay / destroy rounds.

    python tools/diag/graph_soak.py [rounds=200] [model=darknet19]

One run of the round-5 GPU suite died with a host-side segmentation fault inside vt_graph_launch (hipGraphLaunch), in the
captured form of a Darknet-19 f32 step; five re-runs passed.  This loop repeats exactly that -- TrainStep(use_graphs=True),
three replays, a validation step, then the object (and its three-plus graphs) is dropped while nothing is in flight -- with
faulthandler armed, so that a fault leaves the Python frame and the round number behind."""
import faulthandler
import gc
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
faulthandler.enable(all_threads=True)

import torch

from oracle import filler
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    name = sys.argv[2] if len(sys.argv) > 2 else "darknet19"
    ncls, B, S = 24, 12, 96
    x, y = filler.images(B, S).cuda(), filler.labels(B, ncls).cuda()
    t0 = time.time()
    first = None
    for r in range(rounds):
        dt = torch.float32 if r % 2 == 0 else torch.bfloat16
        ts = TrainStep(getattr(backbones, name)(), ncls, B, S, dt, lr=1e-3, momentum=0.9, weight_decay=1e-4,
                       label_smoothing=0.1, device="cuda", use_graphs=True)
        filler.fill_module(ts.model, "va.")
        ts.weights_changed()
        for _ in range(3):
            ts.step(x, y)
        loss = ts.loss()
        got = ts.validate(x, y)
        if dt == torch.float32:
            first = loss if first is None else first
            assert abs(loss - first) < 1e-3 * abs(first), (r, loss, first)
        del ts
        gc.collect()
        if r % 20 == 0:
            print(f"round {r}: loss {loss:.5f} val {got['loss']:.5f} ({time.time() - t0:.0f} s)", flush=True)
    torch.cuda.synchronize()
    print(f"GRAPH_SOAK_OK {rounds} rounds of {name} in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
