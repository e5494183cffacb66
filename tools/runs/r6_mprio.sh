#!/bin/bash
# round 6: s_setprio around the MFMA clusters of vt_igemm / vt_igemm_span / vt_igemm_pspan -- parity + per-layer + step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6mprio
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v -e vt_igemm.o -e vt_igemm_span.o -e vt_igemm_pspan.o)
for f in vt_igemm vt_igemm_span vt_igemm_pspan; do
  /opt/rocm/bin/hipcc $FLAGS -DVT_MFMA_SETPRIO=1 -c $CS/$f.hip -o tools/diag/mprio_$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_mprio.so $OTHERS tools/diag/mprio_vt_igemm.o tools/diag/mprio_vt_igemm_span.o tools/diag/mprio_vt_igemm_pspan.o -ldl || exit 1
python -m pytest tests/test_dwconv_gpu.py tests/test_wgrad6_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a "$OUT/tests.log"
VT_AMD_LIB=$ROOT/tools/diag/libvt_mprio.so python -m pytest tests/test_kernels_gpu.py tests/test_span_s2_gpu.py tests/test_pspan_gpu.py tests/test_dgrad_d2s_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a "$OUT/tests.log"
for rep in 1 2 3; do
  for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_mprio.so"; do
    echo -n "[$(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
    echo -n "[vovnet39 $(basename $lib)] " >> "$OUT/step.log"
    VT_AMD_LIB="$lib" timeout -k 10 300 python3 bench.py --model vovnet39 --steps 20 --warmup 6 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
for lib in "$CS/libvt_amd.so" "$ROOT/tools/diag/libvt_mprio.so"; do
  echo "== $(basename $lib)" >> "$OUT/ops.log"
  VT_AMD_LIB="$lib" timeout -k 10 300 python3 tools/profile_ops.py cspdarknet53 256 16 2>&1 | tail -40 >> "$OUT/ops.log" || echo failed >> "$OUT/ops.log"
done
cat "$OUT/ops.log"
