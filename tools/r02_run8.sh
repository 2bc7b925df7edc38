#!/bin/bash
# timing ablations of gcfn_fused5_kernel on the block alone (M = 256000 rows)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gf5_ablation.txt; : > $S
SEPR_GF_KERNEL=3 timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep "rows" | tee -a $S
for v in "" g5a1 g5a2 g5a4 g5a6 g5a8 g5a16 g5a32 g5a63; do
  SEPR_LIB_VARIANT=$v timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep "rows" | tee -a $S
done
