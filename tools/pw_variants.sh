#!/bin/bash
# GPU box: per-op profile of the pointwise kernels for each library variant under build/variants/ (dev A/B runs)
for v in "$@"; do
  echo "== variant $v"
  VT_AMD_LIB=$PWD/build/variants/libvt_$v.so timeout -k 10 300 python tools/profile_ops.py cspdarknet53 256 60 2>&1 | grep -E "pw_|sum " | head -${PW_LINES:-22}
done
