#!/bin/bash
# quick device loop: GCFN / attention block tests, then the B=16 train bench with kernel stats
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "${KSEL:-gcfn_train or ega_train or dropout_contract or unfused}" -p no:cacheprovider 2>&1 | tail -4 | cut -c1-800
prec=${PREC:-bf16x3}
rm -rf $OUT/prof_train_$prec
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$prec -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 16 --precision $prec > $OUT/prof_train_$prec.log 2>&1)
f=$(find $OUT/prof_train_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_$prec.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms/step %.1f launches/step %.0f' % (tot/3e6, sum(int(r['Calls']) for r in rows)/3))
for r in rows[:14]:
    print('   %-78s %7.1f %8.2f ms %8.1f us' % (r['Name'][:78], int(r['Calls'])/3, float(r['TotalDurationNs'])/3e6, float(r['AverageNs'])/1e3))
PY
find $OUT/prof_train_$prec -name "*kernel_trace.csv" -size +20M -delete
for p in bf16x3 bf16; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch 16 --precision $p 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train $p B=16: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f tn avg %.3f ms x %d' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))"
done
