"""Secondary BASELINE.json configurations (GPU box):

    python tools/bench_configs.py [4|5|all]

config 4: VoVNet-39 forward+backward bf16, batch 256 @224 (fused trainer, same step as bench.py)
config 5: Darknet-YOLOv5x get_feature_maps() multi-scale forward, batch 64 @640, bf16, through the
          module API (one autograd.Function call; no_grad and train-mode-with-grad variants)
Prints one JSON line per configuration."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep

FWD_GFLOP = {"vovnet39": 15.530, "yolov5x": 127.599, "cspdarknet53": 9.335}  # SURVEY 8(d), 2 x MACs per image


def timed(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def config4():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ts = TrainStep(backbones.vovnet39(), 1000, 256, 224, torch.bfloat16, lr=0.05, momentum=0.9, weight_decay=2e-5,
                   label_smoothing=0.1, device=dev)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev))
    dt = timed(ts.step, 5, 20)
    print(json.dumps({"config": "4: VoVNet-39 fwd+bwd(+CE+SGD) bf16 B=256 @224", "ms_per_step": round(dt * 1e3, 3),
                      "images_per_sec": round(256 / dt, 1),
                      "tflops": round(3 * FWD_GFLOP["vovnet39"] * 256 / dt / 1e3, 1), "final_loss": round(ts.loss(), 4)}))


def config5():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = backbones.darknet_yolov5x().to(dev).eval()
    x = torch.rand(64, 3, 640, 640, device=dev)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m.get_feature_maps(x)

    maps = fwd()
    shapes = [tuple(t.shape) for t in maps]
    dt = timed(fwd, 3, 10)
    print(json.dumps({"config": "5: Darknet-YOLOv5x get_feature_maps() bf16 B=64 @640 (eval, no_grad)",
                      "ms": round(dt * 1e3, 3), "images_per_sec": round(64 / dt, 1),
                      "tflops": round(FWD_GFLOP["yolov5x"] * 64 / dt / 1e3, 1), "maps": shapes}))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("4", "all"):
        config4()
    if which in ("5", "all"):
        config5()
