#!/bin/bash
# timing ablations of the projection core (micro-benchmark only: ablated builds compute wrong results)
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/abl_summary.txt
for v in "$@"; do
  echo "===== variant [${v:-default}]" | tee -a $OUT/abl_summary.txt
  SEPR_LIB_VARIANT=$v timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/abl_summary.txt
done
