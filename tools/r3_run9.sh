#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python tools/det_check.py 4 0.05 2>&1 | grep -v amdgpu.ids | head -4
for prec in bf16x3; do
rm -rf $OUT/prof_train_$prec
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$prec -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 16 --precision $prec > $OUT/prof_train_$prec.log 2>&1)
f=$(find $OUT/prof_train_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_$prec.csv && head -12 "$f" | cut -c1-140
find $OUT/prof_train_$prec -name "*kernel_trace.csv" -size +20M -delete
done
for prec in bf16x3 bf16; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch 16 --precision $prec 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train $prec B=16: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f tn avg %.3f ms x %d' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))"
done
