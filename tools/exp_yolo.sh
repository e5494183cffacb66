set -e
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv_ or channel_slices" 2>&1 | tail -2
run() { timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
L="128,128,1,1,28 256,256,1,1,14 512,512,1,1,7 64,64,1,1,56 256,128,1,1,28"
for i in 1 2; do for lib in "" tools/diag/libvt_base.so; do echo "lib=${lib:-new}"; 
  if [ -n "$lib" ]; then export VT_AMD_LIB=$PWD/$lib; else unset VT_AMD_LIB; fi
  VT_BENCH_BATCH=256 VT_BENCH_RESIDUAL=1 timeout -k 10 120 python tools/bench_conv.py fwd $L 2>&1 | grep -v "variant\|amdgpu.ids" | cut -c1-100
  run; done; done
