set -e
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "yolov5_stem" 2>&1 | tail -15
for i in 1 2; do for on in 1 0; do echo "STEM6=$on"; VT_STEM6_KERNEL=$on VT_BENCH_BATCH=64 VT_BENCH_AFFINE=1 timeout -k 10 120 python tools/bench_conv.py fwd 8,80,6,2,640 2>&1 | grep -v "variant\|amdgpu.ids"; done; done
