#!/bin/bash
# round 6: gemm_tn16_kernel - contraction / bf16 training tests, then the in-call A/B of the switch on the bf16 training step
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -q -x -p no:cacheprovider -k "wgrad or bf16 or tn16 or contraction" 2>&1 | tail -4
AB="SEPR_TN16=0 SEPR_TN16=1 SEPR_TN16=0 SEPR_TN16=1" bash tools/r6_train_ab.sh
