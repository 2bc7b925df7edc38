#!/bin/bash
# round-end validation in ONE gpurun call (~6 GPU-minutes): full inference check (incl. Large), all training tests, training bench + profile, 2-rank shared-GPU bench
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LARGE=1 STEPS=10 bash tools/gpu_check.sh > $OUT/gpu_check.log 2>&1
cut -c1-300 $OUT/summary.txt | head -46
bash tools/train_check.sh > $OUT/train_check.log 2>&1
cut -c1-400 $OUT/train_summary.txt | tail -12
for b in 8 16; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b 2>/dev/null | tee $OUT/train_bench_b$b.json | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train B=$b: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f tn avg %.3f ms x %d' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a $OUT/train_summary.txt
done
rm -rf $OUT/prof_train
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 8 > $OUT/prof_train.log 2>&1)
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 | tee -a $OUT/train_summary.txt
find $OUT/prof_train -name "*kernel_trace.csv" -size +20M -delete
echo "== 2 ranks on one GPU (self-launch)" | tee -a $OUT/train_summary.txt
timeout 240 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision 2>&1 | tail -1 | cut -c1-400 | tee -a $OUT/train_summary.txt
