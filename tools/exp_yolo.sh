set -e
run() { timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do for g in 512 256 320; do echo "MINC=$g"; export VT_SPAN_GEMM_MINC=$g; run; timeout -k 10 200 python tools/bench_configs.py 4 2>&1 | grep -o '"ms_per_step": [0-9.]*'; timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -o '"ms": [0-9.]*'; done; done
