set -e
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_span6_gpu.py -m gpu -x -q 2>&1 | tail -3
L64="160,160,3,1,80 320,320,3,1,40 640,640,3,1,20 80,80,3,1,160 80,80,1,1,160"
for i in 1 2; do for lib in "" tools/diag/libvt_u2.so; do echo "lib=${lib:-new}"; 
  if [ -n "$lib" ]; then export VT_AMD_LIB=$PWD/$lib; else unset VT_AMD_LIB; fi
  VT_BENCH_BATCH=64 VT_BENCH_AFFINE=1 VT_BENCH_RESIDUAL=1 timeout -k 10 120 python tools/bench_conv.py fwd $L64 2>&1 | grep -v "variant\|amdgpu.ids" | cut -c1-75
done; done
unset VT_AMD_LIB
for i in 1 2; do for lib in "" tools/diag/libvt_u2.so; do echo "lib=${lib:-new}"; 
  if [ -n "$lib" ]; then export VT_AMD_LIB=$PWD/$lib; else unset VT_AMD_LIB; fi
  timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -v amdgpu.ids | cut -c1-120; done; done
