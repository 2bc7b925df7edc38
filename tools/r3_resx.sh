#!/bin/bash
# GCFN residual-from-planes experiment (SEPR_GF3_RESX): A/B/A/B of the product library and the variant in one call
export TMPDIR=/tmp
for rep in 1 2; do for v in "" resx; do
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[${v:-product}]: %.1f utt/s  %.2f ms/step  parity %.2f dB  pit delta %.2e  gcfn avg %.2f us over %d launches' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['pit_si_snr_max_abs_delta_db'], 1e3*r['roofline']['avg_launch_ms'], r['roofline']['launches']))"
done; done
SEPR_LIB_VARIANT=resx timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gcfn or e2e_golden" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
