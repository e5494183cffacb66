import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from vision_toolbox import _native as N
L = N.lib()
vp = lambda t: C.c_void_p(t.data_ptr())
st = int(torch.cuda.current_stream().cuda_stream)
for dt, tdt in ((N.VT_F32, torch.float32), (N.VT_BF16, torch.bfloat16)):
    for M, Cc in ((512, 64), (512, 32), (2048, 32), (128, 128), (32, 256), (8192, 16), (512, 24)):
        torch.manual_seed(M + Cc)
        dy = torch.randn(M, Cc, device="cuda").to(tdt)
        z = torch.randn(M, Cc, device="cuda").to(tdt)
        co = [torch.randn(Cc, device="cuda") for _ in range(4)]
        ref = None
        bad = 0
        for it in range(300):
            sums = N.stats_buffer(Cc)
            N.check(L.vt_bn_act_bwd_reduce(vp(dy), Cc, vp(z), Cc, vp(co[0]), vp(co[1]), vp(co[2]), vp(co[3]), M, Cc, 1, dt, vp(sums), st))
            d = N.stats_decode(sums)
            if ref is None:
                ref = d
                want = (dy.double() * ((z.double() * co[0].double() + co[1].double()) > 0)).sum(0)
                print(dt, M, Cc, "vs f64:", float((d[0] - want).abs().max() / want.abs().max()))
            elif not torch.equal(d, ref):
                bad += 1
                if bad < 3:
                    print("   MISMATCH it", it, "row0 maxdiff", float((d[0] - ref[0]).abs().max()), "row1", float((d[1] - ref[1]).abs().max()))
        print(dt, M, Cc, "mismatches", bad, flush=True)
