#!/bin/bash
# round 3, call 19: LDS-staged decoder kernel: parity tests, A/B against the direct-from-global kernel in one call
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_infer.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -10 | cut -c1-600
for d in 1 0 1 0; do
  SEPR_DEC_DIRECT=$d timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision --pmc off 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('direct=$d: %.1f utt/s %.3f ms/step parity %.1f dB pit %.2e' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['pit_si_snr_max_abs_delta_db']))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o d -- python $OUT/../bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pd/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'decoder' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, 'us avg', float(r['MaxNs'])/1e3 if 'MaxNs' in r else '')
PY
