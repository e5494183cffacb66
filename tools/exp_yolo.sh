set -e
run() { timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do for lib in "" tools/diag/libvt_base.so; do echo "lib=${lib:-new}"; 
  if [ -n "$lib" ]; then export VT_AMD_LIB=$PWD/$lib; else unset VT_AMD_LIB; fi
  run; timeout -k 10 200 python tools/bench_configs.py 4 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done; done
unset VT_AMD_LIB
timeout -k 10 300 python -m pytest tests/test_span6_gpu.py -m gpu -x -q 2>&1 | tail -2
