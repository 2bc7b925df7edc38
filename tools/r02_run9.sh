#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gf5_variants.txt; : > $S
for v in "" g5p1 g5p2 g5p3 g5ds g5rd3 g5rd1; do
  SEPR_LIB_VARIANT=$v timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep "rows" | tee -a $S
done
for v in gf5acc g5dsacc; do
  SEPR_LIB_VARIANT=$v timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep -v amdgpu.ids | tee -a $S
done
