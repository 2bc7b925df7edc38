#!/bin/bash
# round 6: riding reductions - training tests that compare modes bit for bit, then the in-call A/B on the bf16 training step
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -q -x -p no:cacheprovider -k "deferred or captured or wgrad or tiny or graph_replay or tn16" 2>&1 | tail -4
AB="SEPR_TRAIN_RIDE=0 SEPR_TRAIN_RIDE=1 SEPR_TRAIN_RIDE=0 SEPR_TRAIN_RIDE=1" bash tools/r6_train_ab.sh
