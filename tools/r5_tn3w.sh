#!/bin/bash
# round 5: plain-bf16 contraction with two LDS planes (default build) and, as variant tn3w, compiled for 3 workgroups per CU
export TMPDIR=/tmp
O=gpurun_out/r05_tn3w; mkdir -p $O
K="wgrad or general_loader"
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "$K" 2>&1 | tail -2 | tee $O/tests_default.txt
SEPR_LIB_VARIANT=tn3w SEPR_TN_WGS=768 timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "$K" 2>&1 | tail -2 | tee $O/tests_tn3w.txt
run() { # $1 variant, $2 wgs
  SEPR_LIB_VARIANT=$1 SEPR_TN_WGS=$2 timeout 200 python bench.py --mode train --precision bf16 --batch 16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train bf16 B=16 [lib=${1:-default} wgs=${2:-512}]: %.1f utt/s  %.2f ms/step  loss %.4f' % (r['value'], r['ms_per_step'], r.get('loss', float('nan'))))"
}
{ run "" ""; run tn3w 768; run tn3w ""; run "" ""; run tn3w 768; } 2>&1 | tee $O/ab.txt
