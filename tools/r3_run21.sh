#!/bin/bash
# round 3, call 21: whole-step capture (train_step.CapturedTrainStep): test, bench line in the three launch modes
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "captured_whole_step" -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |Error" | head -20 | cut -c1-1200
for mode in step split; do for p in bf16x3 bf16; do
  timeout 400 python bench.py --mode train --steps 4 --warmup 2 --batch 16 --precision $p --train-graphs $mode 2>$OUT/train_$mode.err | grep '^{' > $OUT/train_${mode}_$p.json
  python - <<PY
import json
try:
    r = json.load(open("$OUT/train_${mode}_$p.json"))
    print('mode=$mode $p B=16: %.1f utt/s %.1f ms/step host_enq %.1f host_loop %.1f loss %.3f gn %.2f fallback=%s' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['host_loop_ms_per_step'], r['loss'], r['grad_norm'], r.get('capture_fallback')))
except Exception as e:
    print('mode=$mode $p: no line', e); print(open("$OUT/train_$mode.err").read()[-1500:])
PY
done; done
