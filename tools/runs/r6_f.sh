#!/bin/bash
# round 6, GPU call E: span6 prologue via LDS staging + fused dgrad/BN-reduce -- bit-equality tests, phase stamps old/new, alternating A/B of the shipped builds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6f
mkdir -p "$OUT"
cd "$ROOT"
CS=$ROOT/vision-toolbox_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I$CS -I$ROOT/include"
OTHERS=$(ls "$CS"/*.o | grep -v vt_igemm_span6.o)
# old (round-5) span6: shipped form and stamped form; new: stamped form
( /opt/rocm/bin/hipcc $FLAGS -c tools/diag/span6_r5.hip -o tools/diag/span6_r5.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_r5.so $OTHERS tools/diag/span6_r5.o -ldl ) &
( /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_DIAG -c tools/diag/span6_r5.hip -o tools/diag/span6_r5d.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_r5d.so $OTHERS tools/diag/span6_r5d.o -ldl ) &
( /opt/rocm/bin/hipcc $FLAGS -DVT_SPAN6_DIAG -c $CS/vt_igemm_span6.hip -o tools/diag/span6_r6d.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libvt_r6d.so $OTHERS tools/diag/span6_r6d.o -ldl ) &
python -m pytest tests/test_span6_gpu.py tests/test_dgrad_bnred_gpu.py -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?; echo "span6 tests exit $rc" | tee -a "$OUT/status.txt"; tail -3 "$OUT/tests.log"
wait
ls -la tools/diag/*.so | tee -a "$OUT/status.txt"
[ $rc -ne 0 ] && exit 1
python -m pytest tests/test_trainer_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "frozen_bn or first_step or grouped or block_vs_reference or train_steps_f32 or bf16_train_step" > "$OUT/tests2.log" 2>&1
rc=$?; echo "model tests exit $rc" | tee -a "$OUT/status.txt"; tail -3 "$OUT/tests2.log"
for lib in r6d; do
  for spec in "256 128,128,3,1,28" "256 256,256,3,1,14" "256 512,512,3,1,7" "128 128,128,3,1,28"; do
    set -- $spec
    echo "### $lib batch $1 layer $2" >> "$OUT/stamps.log"
    VT_AMD_LIB="$ROOT/tools/diag/libvt_$lib.so" VT_SPAN6_ABL=16 VT_BENCH_BATCH=$1 timeout -k 10 120 python3 tools/bench_conv.py fwd $2 >> "$OUT/stamps.log" 2>&1
  done
done
grep -E "###|span6 stamps, us|prologue stamps" "$OUT/stamps.log" | cut -c1-400
LAYERS="128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 128,128,3,1,56 160,160,3,1,28"
for rep in 1 2 3; do
  for lib in "$ROOT/tools/diag/libvt_r5.so" "$CS/libvt_amd.so"; do
    echo "== $lib (round $rep)" >> "$OUT/ab.log"
    VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py fwd $LAYERS 2>&1 | grep GF >> "$OUT/ab.log"
    VT_BENCH_RESIDUAL=1 VT_AMD_LIB="$lib" timeout -k 10 200 python3 tools/bench_conv.py fwd 128,128,3,1,28 256,256,3,1,14 2>&1 | grep GF | sed 's/^/[+res] /' >> "$OUT/ab.log"
  done
done
cat "$OUT/ab.log"
for rep in 1 2; do
  timeout -k 10 200 python3 tools/bench_conv.py bnred 128,128,3,1,28 256,256,3,1,14 512,512,3,1,7 2>&1 | grep GF >> "$OUT/bnred.log"
done
cat "$OUT/bnred.log"
for rep in 1 2 3; do
  for cfg in "VT_FUSE_BNRED=0 VT_AMD_LIB=$ROOT/tools/diag/libvt_r5.so" "VT_FUSE_BNRED=0" "VT_FUSE_BNRED=1"; do
    echo -n "[$cfg] " >> "$OUT/step.log"
    env $cfg timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-secondary --steps-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> "$OUT/step.log" || echo failed >> "$OUT/step.log"
  done
done
cat "$OUT/step.log"
