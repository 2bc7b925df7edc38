#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gf5_sizes.txt; : > $S
for nt in "64 4000" "32 4000" "64 2000" "32 2000" "64 1000" "32 1000" "64 500"; do
  for k in 3 5; do
    SEPR_GF_KERNEL=$k timeout 120 python tools/gf5_trace.py $nt 2>&1 | grep "rows" | tee -a $S
  done
done
SEPR_LIB_VARIANT=gf5acc timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep -v amdgpu.ids | tee -a $S
