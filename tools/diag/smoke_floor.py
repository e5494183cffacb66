import sys, torch
sys.path[:0]=["/root/repo/vision-toolbox_amd","/root/repo"]
from oracle import filler
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep
name="cspdarknet53"
x, y = filler.images(2, 64), filler.labels(2, 16)
gaps=[]; vgaps=[]
for k in range(8):
    pre=f"smoke{k}." if k else "smoke."
    out={}
    for dt in (torch.float32, torch.bfloat16):
        ts = TrainStep(getattr(backbones, name)(), 16, 2, 64, dt, lr=0.0, device="cuda:0")
        filler.fill_module(ts.model, pre); ts.weights_changed()
        v = ts.validate(x.cuda(), y.cuda())["loss"]
        ts.step(x.cuda(), y.cuda())
        out[dt]=(ts.loss(), v)
        del ts
    g=(out[torch.bfloat16][0]-out[torch.float32][0])/out[torch.float32][0]
    vg=(out[torch.bfloat16][1]-out[torch.float32][1])/out[torch.float32][1]
    gaps.append(round(g,4)); vgaps.append(round(vg,5))
print("train-mode loss gap bf16 vs f32 fused, 8 weight fills:", gaps)
print("eval-mode (validate) loss gap:", vgaps)
