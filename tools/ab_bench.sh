#!/bin/bash
# A/B of library builds in ONE gpurun call: projection micro-benchmark + whole-model bench per variant.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/ab_bench.sh "" p1l4 p0l4'
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/ab_summary.txt
for v in "$@"; do
  echo "===== variant [${v:-default}]" | tee -a $OUT/ab_summary.txt
  SEPR_LIB_VARIANT=$v timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_summary.txt
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('  bench: %.1f utt/s  %.2f ms/step  parity %.1f dB  gcfn_up %.1f TF (%.3f ms avg)' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['roofline']['achieved'], r['roofline']['avg_launch_ms']))" | tee -a $OUT/ab_summary.txt
done
