#!/bin/bash
# round 6, last session: GCFN backward middle kernel - where hipcc put its vmcnt(0) (tools/isa_trace.py) and the per-tile constant fetches: GCFN training tests under
# the candidate variants, CLA tests under the product (SEPR_DWWG_NC default 8), then product / top-wait / + constants in LDS / + one barrier on the bf16 training
# step in one call, with the kernel's rocprofv3 average
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for v in ${TEST_VARIANTS:-gbconst gbconstone}; do
  echo "== tests under $v"; SEPR_LIB_VARIANT=$v timeout 600 python -m pytest tests/test_train_gpu.py -q -x -p no:cacheprovider -k "gcfn" 2>&1 | tail -2
done
echo "== CLA tests, product"; timeout 600 python -m pytest tests/test_train_gpu.py -q -x -p no:cacheprovider -k "cla or CLA or dwconv" 2>&1 | tail -2
AB="${AB:-base:|topwait:SEPR_LIB_VARIANT=gbtopwait|const:SEPR_LIB_VARIANT=gbconst|constone:SEPR_LIB_VARIANT=gbconstone}" REPS="2" STEPS=6 PROF="gcfn_bwd_mid" bash tools/r4_ab_train.sh 2>&1 | tee $OUT/r6_gbwait_ab.txt
