#!/bin/bash
# round 6: register epilogue of gcfn_bwd_mid_kernel - GCFN training tests, then the bf16 training step
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -q -x -p no:cacheprovider -k "gcfn" 2>&1 | tail -4
AB="${AB:-SEPR_TN16=1 SEPR_TN16=1}" bash tools/r6_train_ab.sh
