#!/bin/bash
# round 3, call 27: weight-gradient work of the fused GCFN backward on a second stream (SEPR_TRAIN_SIDE=1, experimental): parity /
# determinism subset with the switch on, then the training line with and without it
set -u
export TMPDIR=/tmp
SEPR_TRAIN_SIDE=1 timeout 500 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "gcfn_train or train_step_tiny or full_size or captured_whole_step or train_graph" -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -6 | cut -c1-700
for side in 0 1 0 1; do
  SEPR_TRAIN_SIDE=$side timeout 300 python bench.py --mode train --steps 4 --warmup 2 --batch 16 --precision bf16x3 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('side=$side: %.1f utt/s %.2f ms/step loss %.3f fallback=%s' % (r['value'], r['ms_per_step'], r['loss'], r.get('capture_fallback')))"
done
