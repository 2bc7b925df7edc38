#!/bin/bash
# Large_DM_WHAMR bench A/B of environment settings / library variants in one call (parity comes with the bench line):  AB="SEPR_X=0 SEPR_LIB_VARIANT=xwred" bash tools/r6_large_ab.sh
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for e in ${AB:-SEPR_X=0 SEPR_LIB_VARIANT=xwred SEPR_X=0 SEPR_LIB_VARIANT=xwred}; do
  env $e timeout 300 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 8 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off 2>$OUT/r6_large_err.log | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$e: %.1f utt/s  %.2f ms/step  parity %s dB  pit_delta %s' % (r['value'], r['ms_per_step'], r.get('parity_db_vs_golden'), r.get('pit_si_snr_max_abs_delta_db')))"
done 2>&1 | tee $OUT/r6_large_ab.txt
