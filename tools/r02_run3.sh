#!/bin/bash
# round-2 GPU call 3: training tests (all groups), training profile, fused-GCFN stagger A/B, then the full inference check
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/train_check.sh
echo "== train bench B=4 / B=8" | tee -a $OUT/train_summary.txt
for b in 4 8; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b 2>/dev/null | tee $OUT/train_bench_b$b.json | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train B=$b: %.1f utt/s %.1f ms/step loss %.3f gn %.2f tn avg %.3f ms x %d' % (r['value'], r['ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a $OUT/train_summary.txt
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 4 > $OUT/prof_train.log 2>&1)
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-180 | tee -a $OUT/train_summary.txt
find $OUT/prof_train -name "*kernel_trace.csv" -size +20M -delete
echo "== GCFN stagger A/B" | tee -a $OUT/train_summary.txt
bash tools/ab_env.sh - "SEPR_GF_STAGGER=400" "SEPR_GF_STAGGER=1200" "SEPR_GF_STAGGER=2400" "SEPR_GF_STAGGER=4800" -
cat $OUT/ab_env.txt | tee -a $OUT/train_summary.txt
LARGE=1 STEPS=10 bash tools/gpu_check.sh > $OUT/gpu_check.log 2>&1
cat $OUT/summary.txt | cut -c1-400 | head -60
