"""Probe (GPU box): what a collective library's channel kernels would cost the train step on ONE GPU.

    python tools/rccl_hog.py [batch=128] [steps=20]

During every step a dummy kernel of W persistent workgroups (256 threads, 64 KiB of LDS each, idle until a wall-clock
deadline: vt_debug_hog) sits on a third stream for (a) the whole step and (b) a 1.5 ms window in the middle of backward --
about what the 104 MiB gradient exchange takes on xGMI -- for W in {0, 16, 32}.  The CU-owning kernels (span6, pspan,
wgrad6: one workgroup per CU, static partitions) cannot share a CU with such a workgroup; the JSON line says what that costs."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ts = TrainStep(backbones.cspdarknet53(), 1000, B, 224, torch.bfloat16, lr=0.05, momentum=0.9, weight_decay=2e-5,
                   label_smoothing=0.1, device=dev)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev))
    lib = N.lib()
    # HIP maps streams onto a few hardware queues; a hog that shares the main stream's queue would simply serialise with the
    # step.  Candidates are tried with a ONE-workgroup, LDS-free hog (which costs nothing when it truly runs beside the
    # step): the first stream on which the step keeps its time is used.
    cands = [torch.cuda.Stream(device=dev, priority=p) for p in (0, 0, 0, 0, -1, -1)]
    hog = cands[0]

    def run(wgs, us, delay_us):
        nonlocal hog
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if wgs:
                # the hog starts with the step (ordered behind the previous step's end) and, for the windowed form, idles
                # `delay_us` on ONE workgroup first so that the W-wide part falls into backward
                hog.wait_stream(torch.cuda.current_stream())
                if delay_us:
                    N.check(lib.vt_debug_hog(1, 0, float(delay_us), hog.cuda_stream))
                N.check(lib.vt_debug_hog(wgs, 64 * 1024, float(us), hog.cuda_stream))
            ts.step()
            torch.cuda.current_stream().wait_stream(hog)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    base = run(0, 0, 0)
    probe = {}
    for i, c in enumerate(cands):
        hog = c
        ms = run(1, base * 1e3 * 0.9, 0)
        probe[i] = round(ms, 3)
        if ms < base * 1.03:
            break
    out = {"batch": B, "steps": steps, "step_ms_no_hog": round(base, 3), "one_workgroup_hog_by_candidate_stream": probe,
           "whole_step": {}, "window_1p5ms_in_backward": {}}
    for w in (16, 32):
        ms = run(w, base * 1e3 * 0.97, 0)
        out["whole_step"][str(w)] = {"step_ms": round(ms, 3), "slowdown": round(ms / base - 1, 4)}
        ms = run(w, 1500.0, base * 1e3 * 0.55)
        out["window_1p5ms_in_backward"][str(w)] = {"step_ms": round(ms, 3), "slowdown": round(ms / base - 1, 4)}
    out["step_ms_no_hog_again"] = round(run(0, 0, 0), 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
