#!/bin/bash
# round 3, call 20: decoder with more loads in flight per wave (SEPR_DEC_ILP build) against the product, alternating in one call
set -u
export TMPDIR=/tmp
for v in "" decilp "" decilp; do
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision --pmc off 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('variant=[$v]: %.1f utt/s %.3f ms/step parity %.1f dB pit %.2e' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['pit_si_snr_max_abs_delta_db']))"
done
