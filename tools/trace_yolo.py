"""Dev (GPU box): per-launch table of one Darknet-YOLOv5x get_feature_maps() forward (batch 64 @640, bf16) from a
rocprofv3 --kernel-trace CSV.   rocprofv3 --kernel-trace -d DIR -- python tools/trace_yolo.py run ; python tools/trace_yolo.py show DIR"""
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]


def run():
    import torch
    from vision_toolbox import backbones
    m = backbones.darknet_yolov5x().to("cuda").eval()
    x = torch.rand(64, 3, 640, 640, device="cuda")
    for _ in range(4):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            m.get_feature_maps(x)
    torch.cuda.synchronize()


def show(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # last forward: from the last nchw_to_nhwc launch on
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r["Kernel_Name"]]
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = 0.0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
        tot += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4}  {name[:90]}")
    print(f"{len(rows)} launches, busy {tot / 1e3:.3f} ms, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
