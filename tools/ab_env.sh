#!/bin/bash
# whole-model A/B over environment settings (one bench run each; "-" = defaults):
#   bash tools/ab_env.sh - "SEPR_X3_GRID=tiles" "SEPR_X3_GRID=4 SEPR_GF_GRID=tiles"
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/ab_env.txt
for v in "$@"; do
  [ "$v" = "-" ] && e="" || e="$v"
  env $e timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[$v]: %.1f utt/s  %.2f ms/step  parity %.1f dB  gcfn avg %.3f ms over %d launches' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a gpurun_out/ab_env.txt
done
