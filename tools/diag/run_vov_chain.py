import sys
sys.path[:0]=["/root/repo/tests","/root/repo/vision-toolbox_amd","/root/repo"]
import os
os.environ.setdefault("VT_PW_MIN_MB","0")
import torch, torch.nn.functional as F
import test_fullsize_gpu as T
from vision_toolbox.backbones.vovnet import VoVNet
torch.manual_seed(41)
m = VoVNet(64, [(1, 64, 3, 128), (1, 80, 3, 256)], ese=False)
x = torch.rand(T.B, 3, 112, 112, generator=torch.Generator().manual_seed(42)).to(torch.bfloat16).float()
def ref(x, p, store):
    st = T._st if store else (lambda t: t)
    h = T._ref_unit(x, p["stem.0"], 3, 2, store=store)
    h = T._ref_unit(h, p["stem.1"], 3, 1, store=store)
    h = T._ref_unit(h, p["stem.2"], 3, 1, store=store)
    for si in range(2):
        h = st(F.max_pool2d(h, 3, 2, 1))
        feats = [h]
        for i in range(3):
            feats.append(T._ref_unit(feats[-1], p[f"stages.{si}.module_0.convs.{i}"], 3, 1, store=store))
        h = T._ref_unit(torch.cat(feats, 1), p[f"stages.{si}.module_0.out_conv"], 1, 1, store=store)
    return h
import sys as _s
errs = T._block_case(m, x, ref, torch.float32 if "f32" in _s.argv else torch.bfloat16)
for k,(rel,slope,n,floor) in errs.items():
    print(f"{k:40s} rel {rel:.4f} slope-1 {slope-1:+.5f} n {n:9d} floor {floor:.4f}")
