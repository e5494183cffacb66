set -e
L64="160,320,3,2,160 320,640,3,2,80 640,1280,3,2,40"
for i in 1 2; do for b in 1 2; do echo "BN80=$b";
  VT_IGEMM_BN80=$b VT_BENCH_BATCH=64 VT_BENCH_AFFINE=1 timeout -k 10 120 python tools/bench_conv.py fwd $L64 2>&1 | grep -v "variant\|amdgpu.ids" | cut -c1-110
done; done
