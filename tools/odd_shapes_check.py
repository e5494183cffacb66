"""Odd input shapes through the module API against the oracle (f32 kernels, train + eval)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from oracle import filler, torch_ref as R
from vision_toolbox import backbones

names = ["darknet19", "cspdarknet53", "vovnet27_slim", "darknet_yolov5s", "vovnet39"]
shapes = [(2, 3, 75, 91), (3, 3, 160, 96), (1, 3, 33, 47), (2, 3, 130, 130), (5, 3, 64, 200)]
bad = 0
for name in names:
    for shp in shapes:
        for training in (True, False):
            try:
                m = getattr(backbones, name)()
                filler.fill_module(m, f"odd.{name}.")
                sd = {k: v.clone() for k, v in m.state_dict().items()}
                x = filler.tensor(f"odd{shp}", shp)
                ref = R.feature_maps(name, {k: v.clone() for k, v in sd.items()}, x, training)
                m = m.cuda().train(training)
                with torch.no_grad():
                    maps = m.get_feature_maps(x.cuda())
                torch.cuda.synchronize()
                if ref is None:
                    print(name, shp, training, [tuple(t.shape) for t in maps]); continue
                errs = [float((a.cpu() - b).norm() / (b.norm() + 1e-12)) for a, b in zip(maps, ref)]
                ok = all(e < 2e-3 for e in errs) and all(tuple(a.shape) == tuple(b.shape) for a, b in zip(maps, ref))
                print("OK " if ok else "BAD", name, shp, "train" if training else "eval", ["%.1e" % e for e in errs], flush=True)
                bad += not ok
            except Exception as e:  # noqa
                print("EXC", name, shp, training, repr(e)[:300], flush=True)
                bad += 1
print("bad:", bad)
