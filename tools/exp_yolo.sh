set -e
timeout -k 10 280 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "80_wide or conv_forward_and_stats or channel_slices" 2>&1 | tail -3
for i in 1 2; do for b in 0 1; do echo "BN80=$b"; VT_IGEMM_BN80=$b timeout -k 10 200 python tools/bench_configs.py 5 2>&1 | grep -v amdgpu.ids; done; done
