#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/run14_summary.txt; : > $S
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocks or e2e_golden or bit_identical or batch or ragged or pit" 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee -a $S
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('  bench: %.1f utt/s  %.2f ms/step  parity %.1f dB  pit %.1e  gcfn %.1f TF algo (%.3f ms avg x %d)' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r.get('pit_si_snr_max_abs_delta_db', -1), r['roofline']['achieved'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a $S
done
bash tools/prof_variants.sh "" 2>&1 | tail -18 | tee -a $S
