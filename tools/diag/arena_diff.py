"""find the first buffer (in build order) that differs between a majority-class run and a deviating run"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]
import torch
from vision_toolbox import backbones
from vision_toolbox import engine as E
from vision_toolbox.trainer import TrainStep
dt = torch.float32
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 3, 64, 64, generator=g).cuda(); y = torch.randint(0, 16, (8,), generator=g).cuda()
runs = []
for r in range(30):
    torch.manual_seed(0)
    ts = TrainStep(backbones.darknet_yolov5n(), 16, 8, 64, dt, lr=0.0, use_graphs=False)
    ts.step(x, y); torch.cuda.synchronize()
    runs.append((ts.gflat.double().clone(), ts.arena.clone()))
    if r == 0:
        b = ts.prog.builder
        refs = []
        for name, t in b.debug_refs.items():
            refs.append((t.buf.offset, t.buf.nbytes, name, "act"))
            gs = b.gstate.get(id(t.buf))
            if gs is not None and gs.gbuf is not None:
                refs.append((gs.gbuf.offset, gs.gbuf.nbytes, name, "grad"))
        zf, zb = ts.prog.zf_off, ts.prog.zb_off
        from vision_toolbox import _native as N
        fins = []
        for phase, ops, n in (("fwd", ts.prog.fwd_ops, ts.prog.n_fwd), ("bwd", ts.prog.bwd_ops, ts.prog.n_bwd)):
            for idx in range(n):
                op = ops[idx]
                k = op.kind & 0xFFFF
                if k in (N.OP_BN_FINALIZE, N.OP_BN_BWD_FINALIZE):
                    base = {E.ZERO_F: zf, E.ZERO_B: zb}[op.ptr[0].base]
                    cpos = 6 if k == N.OP_BN_BWD_FINALIZE else 6
                    fins.append((phase, idx, base + op.ptr[0].offset, op.i[0], op.ptr[cpos].offset, op.tag))
        print("arena", ts.arena.numel(), "zf_off", zf, "zb_off", zb, "n refs", len(refs))
    del ts
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
bad = [i for i in range(1, len(runs)) if rel(runs[i][0], runs[0][0]) > 1e-4]
print("deviating runs:", bad)
if bad:
    # is run 0 in the majority?
    good = 0 if len(bad) < len(runs) / 2 else bad[0]
    dev = bad[0] if good == 0 else 0
    A, B = runs[good][1], runs[dev][1]
    out = []
    for off, nb, name, kind in refs:
        a = A[off:off + nb].view(torch.float32).double(); b_ = B[off:off + nb].view(torch.float32).double()
        out.append((off, name, kind, rel(b_, a)))
    out.sort()
    for off, name, kind, d in out:
        if kind == "grad" or name.endswith(".dz"):
            print(f"{off:10d} {kind:4s} {name:40s} {d:.2e}")
    from vision_toolbox import _native as N
    for phase, idx, off, Cc, coff, tag in fins:
        nb = N.stat_floats(Cc) * 4
        a = A[off:off + nb].view(torch.int64).view(32, 2, Cc, 2); b_ = B[off:off + nb].view(torch.int64).view(32, 2, Cc, 2)
        ta, tb = N.stats_decode(a), N.stats_decode(b_)
        raw = int((a != b_).sum())
        ncoef = (3 if phase == "bwd" else 4) * Cc * 4
        ca = A[coff:coff + ncoef].view(torch.float32).double(); cb = B[coff:coff + ncoef].view(torch.float32).double()
        d0, d1 = rel(tb[0], ta[0]), rel(tb[1], ta[1])
        if raw or d0 > 0 or rel(cb, ca) > 0:
            print(f"{phase} op {idx:4d} tag {tag:3d} C {Cc:4d} raw words differing {raw:6d} totals row0 {d0:.2e} row1 {d1:.2e} coef {rel(cb, ca):.2e}")
    # raw statistic regions
    for nm, o, n in (("ZERO_F", zf, zb - zf), ("ZERO_B", zb, A.numel() - zb)):
        a = A[o:o + n].view(torch.int64); b_ = B[o:o + n].view(torch.int64)
        nz = (a != b_).nonzero().flatten()
        print(nm, "int64 words differing:", nz.numel(), "first at byte", (int(nz[0]) * 8 if nz.numel() else None))
