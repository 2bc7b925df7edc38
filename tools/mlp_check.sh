#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/mlp_check.txt; : > $S
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocks or e2e_golden or bit_identical or batch or pit or ragged or intermediate" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 | tee -a $S
for e in SEPR_FUSE_MLP=0 SEPR_FUSE_MLP=1 SEPR_FUSE_MLP=0 SEPR_FUSE_MLP=1; do
env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('  [$e] bench: %.1f utt/s  %.2f ms/step  parity %.1f dB  pit %.1e' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r.get('pit_si_snr_max_abs_delta_db', -1)))" | tee -a $S
done
bash tools/prof_variants.sh "" 2>&1 | tail -17 | tee -a $S
