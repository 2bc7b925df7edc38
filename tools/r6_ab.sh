#!/bin/bash
# round 6: A/B of environment settings inside ONE gpurun call:  bash tools/r6_ab.sh "SEPR_GF_LAT=0 SEPR_FOLD_HEAD=0" "SEPR_GF_LAT=3" ...
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for e in "$@"; do
  echo "== $e"
  env $e timeout 600 python tools/r6_lat.py 2>$OUT/r6_lat_err.log | tail -1 | tee -a $OUT/r6_ab.jsonl
done
