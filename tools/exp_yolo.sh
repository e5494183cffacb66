set -e
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "maxpool or vovnet or pool" 2>&1 | tail -2
for lib in "" tools/diag/libvt_base.so; do echo "lib=${lib:-new}"; 
  if [ -n "$lib" ]; then export VT_AMD_LIB=$PWD/$lib; else unset VT_AMD_LIB; fi
  timeout -k 10 300 python tools/profile_ops.py vovnet39 256 5 2>&1 | grep -E "maxpool|ops, sum"; done
