set -e
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "80_wide" 2>&1 | tail -2
run() { timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do for w8 in 1 3; do echo "W8=$w8"; export VT_IGEMM_W8=$w8; run; timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -o '"ms": [0-9.]*'; done; done
