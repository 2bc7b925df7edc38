#!/bin/bash
# round 3, call 25: the batch as two half-batch pipelines on two streams (SEPR_PIPELINES=2) against one, alternating
set -u
export TMPDIR=/tmp
for pl in 2 3 4 2 4; do
  SEPR_PIPELINES=$pl timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision --pmc off 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('pipelines=$pl: %.1f utt/s %.3f ms/step parity %.1f dB' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden']))"
done
