#!/bin/bash
# round 6: Large_DM_WHAMR - parity of the statistics chain, then bench A/B chain on / off in one call
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "large or Large or wide_core or tiny_s3 or crosses_maxlen" > $OUT/r6_large_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/r6_large_pytest.log
for e in "SEPR_CHAIN_STATS=0" "SEPR_CHAIN_STATS=1" "SEPR_CHAIN_STATS=0" "SEPR_CHAIN_STATS=1"; do
  env $e timeout 600 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off 2>$OUT/r6_large_err.log | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$e: %.1f utt/s  %.2f ms/step  parity %s dB' % (r['value'], r['ms_per_step'], r.get('parity_db_vs_golden')))"
done
