#!/bin/bash
# fast GPU loop: block + e2e parity (default lib), then a bench line per library variant / env setting
#   bash tools/quick_check.sh "" gfv1 "SEPR_X3_GRID=4"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_blocks or test_e2e_golden or test_linear_core or test_full_size" -p no:cacheprovider > $OUT/quick_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/quick_pytest.log
for v in "$@"; do
  if [[ "$v" == *=* ]]; then e="$v"; else e="SEPR_LIB_VARIANT=$v"; fi
  env $e timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[$v]: %.1f utt/s  %.2f ms/step  parity %.1f dB  gcfn avg %.3f ms' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['roofline']['avg_launch_ms']))"
done
