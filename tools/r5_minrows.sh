#!/bin/bash
# round 5: contraction plan, minimum rows per slice (SEPR_TN_MINROWS, default 256) over the whole training step
export TMPDIR=/tmp
O=gpurun_out/r05_minrows; mkdir -p $O
run() {
  SEPR_TN_MINROWS=$1 SEPR_TN_WGS=$2 timeout 200 python bench.py --mode train --precision bf16 --batch 16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train bf16 B=16 [minrows=${1:-256} wgs=${2:-512}]: %.1f utt/s  %.2f ms/step' % (r['value'], r['ms_per_step']))"
}
{ run "" ""; run 128 ""; run 512 ""; run 1024 ""; run "" ""; } 2>&1 | tee $O/ab.txt
