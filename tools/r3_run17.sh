#!/bin/bash
# round 3, call 17: timing ablations of the bf16x3 projection core at the Large (F = 256) shapes
set -u
export TMPDIR=/tmp
export GEMM_SHAPES="256000,1536,256;256000,256,768;256000,768,256;256000,256,256;128000,768,128;8192,8192,4096"
export GEMM_SKIP_F32=1
for v in "" ablmma ablld ablw ablst ablall; do
  SEPR_LIB_VARIANT=$v timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "variant|M=" | cut -c1-200
done
