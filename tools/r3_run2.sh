#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python tools/det_check.py 4 0.05 2>&1 | grep -v amdgpu.ids | tail -70
for prec in bf16x3 bf16; do
rm -rf $OUT/prof_train_$prec
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$prec -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 16 --precision $prec > $OUT/prof_train_$prec.log 2>&1)
f=$(find $OUT/prof_train_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_$prec.csv && head -25 "$f" | cut -c1-160
find $OUT/prof_train_$prec -name "*kernel_trace.csv" -size +20M -delete
done
timeout 300 python tools/train_host_profile.py 16 2>&1 | grep -v amdgpu.ids | head -60
