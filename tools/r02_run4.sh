#!/bin/bash
# round-2 GPU call 4: re-check the training kernels that changed, training bench + profile
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/train_summary.txt
for grp in test_wgrad_core test_gcfn_train test_cla_train test_ega_train test_downconv_split_fuse test_front_and_heads test_train_step_tiny; do
  timeout 420 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "$grp" -p no:cacheprovider > "$OUT/train_$grp.log" 2>&1
  echo "pytest [$grp] rc=$?" | tee -a $OUT/train_summary.txt
  grep -E "passed|failed|error" "$OUT/train_$grp.log" | tail -2 | tee -a $OUT/train_summary.txt
  grep -E "^E  " "$OUT/train_$grp.log" | head -6 | cut -c1-1200 | tee -a $OUT/train_summary.txt
done
for b in 4 8 16; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b 2>/dev/null | tee $OUT/train_bench_b$b.json | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train B=$b: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f tn avg %.3f ms x %d' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))" | tee -a $OUT/train_summary.txt
done
rm -rf $OUT/prof_train
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 8 > $OUT/prof_train.log 2>&1)
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -36 "$f" | cut -c1-150 | tee -a $OUT/train_summary.txt
find $OUT/prof_train -name "*kernel_trace.csv" -size +20M -delete
if [ -n "${EXTRA:-}" ]; then bash -c "$EXTRA" 2>&1 | tail -30 | tee -a $OUT/train_summary.txt; fi
