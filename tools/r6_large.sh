#!/bin/bash
# round 6: Large_DM_WHAMR - parity, then bench A/B of an environment switch (default: the fused F = 256 GCFN off / on) in one call
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "large or Large or wide_core" > $OUT/r6_large_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/r6_large_pytest.log
for e in ${AB:-SEPR_FUSE_GCFN256=0 SEPR_FUSE_GCFN256=1 SEPR_FUSE_GCFN256=0 SEPR_FUSE_GCFN256=1}; do
  env $e timeout 600 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off 2>$OUT/r6_large_err.log | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$e: %.1f utt/s  %.2f ms/step  parity %s dB  pit_delta %s' % (r['value'], r['ms_per_step'], r.get('parity_db_vs_golden'), r.get('pit_si_snr_max_abs_delta_db')))"
done
