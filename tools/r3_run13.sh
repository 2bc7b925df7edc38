#!/bin/bash
# round 3, call 13: timing ablations of the weight-gradient contraction at the GCFN shapes (what bounds it?)
set -u
export TMPDIR=/tmp
for v in "" tn1 tn2 tn3 tn4 tn7; do
  echo "== variant '${v}' (1 no MFMA, 2 no conversion/LDS staging, 4 no global loads)"
  SEPR_LIB_VARIANT=$v timeout 200 python tools/wgrad_bench.py 2>&1 | grep "M=" | head -12
done
