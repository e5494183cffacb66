set -e
timeout -k 10 300 python -m pytest tests/test_pointwise_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do for on in 1 0; do echo "PW_INFERENCE=$on"; VT_PW_INFERENCE=$on timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -o '"ms": [0-9.]*'; done; done
timeout -k 10 500 python -m pytest tests/test_modules_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "yolo or darknet or feature or module" 2>&1 | tail -2
