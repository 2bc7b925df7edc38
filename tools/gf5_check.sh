#!/bin/bash
# the experimental GCFN kernel variants: parity (block tests through SEPR_GF_KERNEL=5), block timing, phase accumulators, bench
#   bash tools/gf5_check.sh gf5 gf5x1 gf5x2        (library variants; build them first)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gf5_check.txt; : > $S
SEPR_GF_KERNEL=3 timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep "rows" | tee -a $S
SEPR_GF_KERNEL=3 timeout 120 python tools/gf5_trace.py 32 4000 2>&1 | grep "rows" | tee -a $S
for v in "$@"; do
  echo "== $v" | tee -a $S
  SEPR_LIB_VARIANT=$v SEPR_GF_KERNEL=5 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gcfn_block_large or bit_identical" 2>&1 | grep -E "passed|failed|Error" | tail -2 | tee -a $S
  SEPR_LIB_VARIANT=$v SEPR_GF_KERNEL=5 timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep -v amdgpu.ids | tee -a $S
  SEPR_LIB_VARIANT=$v SEPR_GF_KERNEL=5 timeout 120 python tools/gf5_trace.py 32 4000 2>&1 | grep "rows" | tee -a $S
done
