#!/bin/bash
# round 6: in-call A/B of environment settings on the bf16 training step (batch 16), then the training tests under the default
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for e in ${AB:-SEPR_TRAIN_WGRAD_STREAM=0 SEPR_TRAIN_WGRAD_STREAM=1 SEPR_TRAIN_WGRAD_STREAM=0 SEPR_TRAIN_WGRAD_STREAM=1}; do
  env $(echo $e | tr , " ") timeout 400 python bench.py --mode train --batch ${BATCH:-16} --steps 6 --warmup 2 --precision ${PRECISION:-bf16} ${EXTRA:-} 2>$OUT/r6_train_ab_err.log | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); print('$e: %.1f utt/s  %.2f ms/step  loss %s gn %s' % (r['value'], r['ms_per_step'], r.get('loss'), r.get('grad_norm')))
except Exception as ex:
    print('$e FAILED', ex)"
done
tail -3 $OUT/r6_train_ab_err.log
