#!/bin/bash
set -u
export TMPDIR=/tmp
for v in "" tns60 tns110; do
  echo "== variant '${v}'"
  SEPR_LIB_VARIANT=$v WGRAD_NORM=1 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "M=" | head -8
done
