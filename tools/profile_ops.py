"""Per-op timing of the CSPDarknet-53 bf16 train step (GPU box).

    python tools/profile_ops.py [model] [batch] [top]

Replays every op of the forward / backward launch lists in isolation (each op 5x back to
back on one stream, HIP events around it) and prints the time, the conv shape and the
achieved TFLOP/s or GB/s, sorted by time, plus totals per op kind.  In isolation = without
the filter-gradient side stream competing, so the column sums are a lower bound of a step."""
import ctypes
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep


# yardstick of the "excess" columns: the chip table of MI355X_MICROARCH.md -- dense bf16 MFMA 2.5 PFLOP/s, HBM 6.3 TB/s
# achievable (8 TB/s spec)
PEAK_FLOPS, PEAK_HBM = 2.5e15, 6.3e12


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "cspdarknet53"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ts = TrainStep(getattr(backbones, model)(), 1000, B, 224, torch.bfloat16, lr=0.05, momentum=0.9,
                   weight_decay=2e-5, label_smoothing=0.1, device=dev)
    ts.images.copy_(torch.rand(ts.images.shape, device=dev))
    ts.labels.copy_(torch.randint(0, 1000, ts.labels.shape, device=dev))
    for _ in range(3):
        ts.step()
    torch.cuda.synchronize()
    s = int(torch.cuda.current_stream().cuda_stream)
    p = ts.prog
    rows = []
    excess = []
    algo_bytes = defaultdict(float)
    by_pix = defaultdict(lambda: [0.0, 0])
    reps = 5
    for phase, ops, n in (("fwd", p.fwd_ops, p.n_fwd), ("bwd", p.bwd_ops, p.n_bwd)):
        group_ms = {}  # op index -> per-layer ms of the grouped filter-gradient run it belongs to
        for idx in range(n):
            op = ops[idx]
            kind = op.kind & 0xFFFF
            if kind in (N.OP_FORK, N.OP_JOIN, N.OP_FORK_MARK, N.OP_FORK_WAIT):
                continue
            cnt = 1
            if kind == N.OP_CONV_WGRAD and idx not in group_ms:
                # a run of same-descriptor filter gradients is ONE grouped call in the step (vt_conv_wgrad_group)
                dsz = ctypes.sizeof(N.ConvDesc)
                while (idx + cnt < n and (ops[idx + cnt].kind & 0xFFFF) == kind
                       and bytes(ops[idx + cnt].i)[:dsz + 4] == bytes(op.i)[:dsz + 4]):
                    cnt += 1
            if idx in group_ms:
                ms = group_ms[idx]
            else:
                one = (N.Op * cnt).from_address(ctypes.addressof(ops) + idx * ctypes.sizeof(N.Op))
                N.run_ops(one, cnt, ts.bases, s)
                e0, e1 = N.Event(), N.Event()
                e0.record(s)
                for _ in range(reps):
                    N.run_ops(one, cnt, ts.bases, s)
                e1.record(s)
                ms = e0.elapsed_ms(e1) / reps / cnt
                for j in range(1, cnt):
                    group_ms[idx + j] = ms
            name = N.OP_NAMES.get(kind, str(kind))
            desc, work = "", ""
            if kind in (N.OP_CONV_IGEMM, N.OP_CONV_WGRAD):
                d = N.ConvDesc.from_buffer_copy(bytes(op.i)[: ctypes.sizeof(N.ConvDesc)])
                fl = 2.0 * d.B * d.Ho * d.Wo * d.Cout * d.Cin * d.ntaps
                desc = f"{d.Cin:4d}->{d.Cout:4d} taps {d.ntaps:2d} s{d.sh} in {d.Hi:3d}x{d.Wi:<3d} grid {d.Ho:3d}x{d.Wo:<3d}"
                nb = 2.0 * (d.B * d.Hi * d.Wi * d.Cin / (d.sh * d.sw if kind == N.OP_CONV_WGRAD else 1)
                            + d.B * d.Ho * d.Wo * d.Cout) + 2.0 * d.Cout * d.Cin * d.ntaps
                if kind == N.OP_CONV_WGRAD:
                    nb = 2.0 * (d.B * d.Hi * d.Wi * d.Cin + d.B * d.Ho * d.Wo * d.Cout)
                ideal = max(fl / PEAK_FLOPS, nb / PEAK_HBM) * 1e3  # ms against the guide's peaks
                work = f"{fl / ms / 1e9:7.1f} TF/s {nb / ms / 1e9:6.2f} TB/s ideal {ideal:6.3f} excess {ms - ideal:6.3f}"
                excess.append((ms - ideal, phase, name, desc, ms, ideal))
            # bytes this decomposition moves when every operand is read / written exactly once (bf16 activations)
            if kind in (N.OP_CONV_IGEMM, N.OP_CONV_WGRAD):
                algo_bytes[name] += nb
            elif kind == N.OP_BN_ACT_APPLY:  # i: ldz ldr ldy C relu dtype | f: M ; ptr 3 = residual
                nb = 2.0 * op.f[0] * op.i[3] * (2 + (1 if op.ptr[3].base >= 0 else 0))
                if op.ptr[5].base >= 0:  # fused max-pool: + pooled map and arg-max bytes
                    nb += (2.0 + 1.0) * op.f[0] * op.i[3] / 4
                algo_bytes[name] += nb
                desc = f"C {op.i[3]:4d} M {int(op.f[0])}{' +res' if op.ptr[3].base >= 0 else ''}{' +pool' if op.ptr[5].base >= 0 else ''}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind == N.OP_BN_FIN_APPLY:  # (round 6) ptr 11 = residual | i: C ldz ldr ldy relu dtype | f: count eps momentum M
                nb = 2.0 * op.f[3] * op.i[0] * (2 + (1 if op.ptr[11].base >= 0 else 0))
                algo_bytes[name] += nb
                desc = f"C {op.i[0]:4d} M {int(op.f[3])}{' +res' if op.ptr[11].base >= 0 else ''}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind == N.OP_BN_BWD_FIN_APPLY:  # (round 6) i: C train lddy ldz lddz relu dtype | f: count pscale M
                nb = 2.0 * op.f[2] * op.i[0] * 3
                algo_bytes[name] += nb
                desc = f"C {op.i[0]:4d} M {int(op.f[2])}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind == N.OP_BN_BWD_REDUCE:  # i: lddy ldz C relu dtype | f: M
                pooled = op.ptr[7].base >= 0
                nb = 2.0 * op.f[0] * op.i[2] * (1.25 + 0.125 if pooled else 2)
                algo_bytes[name] += nb
                desc = f"C {op.i[2]:4d} M {int(op.f[0])}{' pooled-dy' if pooled else ''}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind == N.OP_BN_BWD_APPLY:  # i: lddy ldz lddz C relu dtype | f: M
                pooled = op.ptr[6].base >= 0
                nb = 2.0 * op.f[0] * op.i[3] * (2.25 + 0.125 if pooled else 3)
                algo_bytes[name] += nb
                desc = f"C {op.i[3]:4d} M {int(op.f[0])}{' pooled-dy' if pooled else ''}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind in (N.OP_PW_STATS, N.OP_PW_APPLY, N.OP_PW_REDUCE, N.OP_PW_BWD, N.OP_PW_APPLY_FIN, N.OP_PW_BWD_FIN):
                # i: K ngroups relu C0 C1 ldx ... | f: M.  x once; y / dy once; residual / addend and dz where present
                K, Nn, M_ = op.i[0], op.i[3] + op.i[4], op.f[0]
                nb = 2.0 * M_ * K
                kind = {N.OP_PW_APPLY_FIN: N.OP_PW_APPLY, N.OP_PW_BWD_FIN: N.OP_PW_BWD}.get(kind, kind)  # (same operands)
                if kind == N.OP_PW_APPLY:
                    nb += 2.0 * M_ * Nn + sum(2.0 * M_ * op.i[3 + g] for g in range(2) if op.ptr[6 + g].base >= 0)
                elif kind == N.OP_PW_REDUCE:
                    nb += 2.0 * M_ * Nn
                elif kind == N.OP_PW_BWD:
                    nb += 2.0 * M_ * Nn + 2.0 * M_ * K * (2 if op.ptr[9].base >= 0 else 1)
                    nb += sum(2.0 * M_ * op.i[3 + g] for g in range(2) if op.ptr[12 + g].base >= 0)
                algo_bytes[name] += nb
                desc = f"{K:4d}->{Nn:4d} M {int(M_)}"
                work = f"{nb / ms / 1e9:6.2f} TB/s ideal {nb / PEAK_HBM * 1e3:6.3f} excess {ms - nb / PEAK_HBM * 1e3:6.3f}"
            elif kind == N.OP_STEM_BWD_REDUCE:  # i: dtype B H W C ...
                algo_bytes[name] += 2.0 * op.i[1] * op.i[2] * op.i[3] * (8 + 2 * op.i[4])
            # rows (pixels) the op works on: the stage it belongs to
            if kind in (N.OP_CONV_IGEMM, N.OP_CONV_WGRAD):
                npix = d.B * max(d.Ho * d.Wo, d.oH * d.oW if kind == N.OP_CONV_IGEMM else 0)
            elif kind in (N.OP_BN_ACT_APPLY, N.OP_BN_BWD_REDUCE, N.OP_BN_BWD_APPLY, N.OP_PW_STATS, N.OP_PW_APPLY, N.OP_PW_REDUCE, N.OP_PW_BWD):
                npix = int(op.f[0])
            elif kind in (N.OP_BN_FIN_APPLY, N.OP_BN_BWD_FIN_APPLY):
                npix = int(op.f[3] if kind == N.OP_BN_FIN_APPLY else op.f[2])
            elif kind == N.OP_STEM_BWD_REDUCE:
                npix = op.i[1] * op.i[2] * op.i[3]
            else:
                npix = 0
            by_pix[npix][0] += ms
            by_pix[npix][1] += 1
            rows.append((ms, phase, idx, name, desc, work))
    tot = defaultdict(float)
    for ms, phase, idx, name, desc, work in rows:
        tot[(phase, name)] += ms
    print(f"{model} B={B}: {len(rows)} ops, sum {sum(r[0] for r in rows):.3f} ms (isolated, serial)")
    for (phase, name), v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"  {phase} {name:16s} {v:8.3f} ms")
    print(f"-- bytes per step when every operand of this decomposition moves once: {sum(algo_bytes.values()) / 1e9:.2f} GB "
          + ", ".join(f"{k} {v / 1e9:.2f}" for k, v in sorted(algo_bytes.items(), key=lambda kv: -kv[1])))
    print("-- time by the number of pixels an op works on (0: per-channel / plumbing kernels)")
    for npix, (ms_, cnt) in sorted(by_pix.items(), key=lambda kv: -kv[0]):
        print(f"  {npix:10d} pixels: {cnt:4d} ops {ms_:7.3f} ms")
    print("-- pointwise ops (bytes = operands of the pass, once)")
    for ms, phase, idx, name, desc, work in sorted([r for r in rows if r[3].startswith("pw_")], key=lambda r: -r[0])[:top]:
        print(f"  {phase}[{idx:4d}] {name:11s} {desc}  {ms:7.4f} ms {work}")
    print("-- BatchNorm streaming passes by shape (bytes = operands of the pass, once; against 6.3 TB/s)")
    bagg = defaultdict(lambda: [0, 0.0, 0.0])
    for ms, phase, idx, name, desc, work in rows:
        if name.startswith("bn_") and desc:
            a = bagg[(name, desc)]
            a[0] += 1
            a[1] += ms
            a[2] += float(work.split("ideal")[1].split()[0])
    for (name, desc), (cnt, ms, ideal) in sorted(bagg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:top]:
        print(f"  {name:14s} {desc:34s} x{cnt:2d}  {ms:7.3f} ms ideal {ideal:7.3f} excess {ms - ideal:7.3f}  ({ideal / ms * 6.3:4.2f} TB/s)")
    print("-- conv ops by time")
    conv = [r for r in rows if r[3].startswith("conv")]
    for ms, phase, idx, name, desc, work in sorted(conv, key=lambda r: -r[0])[:top]:
        print(f"  {phase}[{idx:4d}] {name:11s} {desc}  {ms:7.4f} ms {work}")
    print("-- conv ops by excess over max(flops / 2.5 PF/s, bytes / 6.3 TB/s)")
    agg2 = defaultdict(lambda: [0, 0.0, 0.0])
    for ex, phase, name, desc, ms, ideal in excess:
        a = agg2[(phase, name, desc)]
        a[0] += 1
        a[1] += ms
        a[2] += ideal
    for (phase, name, desc), (cnt, ms, ideal) in sorted(agg2.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:top]:
        print(f"  {phase} {name:11s} {desc} x{cnt:2d}  {ms:7.3f} ms ideal {ideal:7.3f} excess {ms - ideal:7.3f}")
    print(f"  total conv excess {sum(e[0] for e in excess):.3f} ms of {sum(e[4] for e in excess):.3f} ms")
    # aggregate identical conv shapes
    agg = defaultdict(lambda: [0, 0.0])
    for ms, phase, idx, name, desc, work in conv:
        a = agg[(phase, name, desc)]
        a[0] += 1
        a[1] += ms
    print("-- conv shapes aggregated")
    for (phase, name, desc), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {phase} {name:11s} {desc} x{cnt:2d}  {ms:7.3f} ms")


if __name__ == "__main__":
    main()
