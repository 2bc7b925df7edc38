#!/bin/bash
# round 3, call 12: XCD-aware tile walk of the weight-gradient contraction (tests, bench, PMC traffic)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "wgrad or gcfn_train or train_step_tiny" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-600
for p in bf16x3 bf16; do
  timeout 300 python bench.py --mode train --steps 4 --warmup 2 --batch 16 --precision $p 2>/dev/null | grep '^{' > $OUT/train_r12_$p.json
  python - <<PY
import json
r = json.load(open("$OUT/train_r12_$p.json"))
print('train $p B=16: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm']), json.dumps(r['roofline'])[:400])
PY
done
bash tools/pmc_train_tn.sh 2>&1 | grep -v rocprofv3 | tail -30
