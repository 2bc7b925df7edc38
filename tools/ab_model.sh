#!/bin/bash
# whole-model A/B over library variants: parity of the fused path + bench
export TMPDIR=/tmp
for v in "$@"; do
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[${v:-default}]: %.1f utt/s  %.2f ms/step  parity %.1f dB  site avg %.3f ms over %d launches' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))"
done
