#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/run15_summary.txt; : > $S
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gcfn or blocks or bit_identical" 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee -a $S
for v in oldg "" oldg ""; do
  SEPR_LIB_VARIANT=$v timeout 120 python tools/gf5_trace.py 64 4000 2>&1 | grep "rows" | tee -a $S
  SEPR_LIB_VARIANT=$v timeout 120 python tools/gf5_trace.py 8 1800 2>&1 | grep "rows" | tee -a $S
done
for v in oldg "" oldg ""; do
SEPR_LIB_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('  [$v] bench: %.1f utt/s  %.2f ms/step  parity %.1f dB  pit %.1e  gcfn %.1f TF algo (%.3f ms avg x %d)' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r.get('pit_si_snr_max_abs_delta_db', -1), r['roofline']['achieved'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a $S
done
