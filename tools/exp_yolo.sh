set -e
timeout -k 10 600 python -m pytest tests/test_span6_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do for on in 1 0; do echo "SPLIT=$on"; VT_SPAN6_SPLIT=$on timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -o '"ms": [0-9.]*'; VT_SPAN6_SPLIT=$on timeout -k 10 200 python tools/bench_configs.py 4 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done; done
