#!/bin/bash
# round 3, call 11: packed two-ring depthwise weight gradient; A/B of the three-workgroups-per-CU GCFN backward middle kernel
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "cla_train or gcfn_train" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-600
SEPR_GB_WPS=3 timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "gcfn_train or train_step_tiny" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-600
for wps in 2 3; do
  for p in bf16x3 bf16; do
    SEPR_GB_WPS=$wps timeout 300 python bench.py --mode train --steps 4 --warmup 2 --batch 16 --precision $p 2>/dev/null | grep '^{' > $OUT/train_wps${wps}_$p.json
    python - <<PY
import json
r = json.load(open("$OUT/train_wps${wps}_$p.json"))
print('WPS=$wps train $p B=16: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm']), json.dumps(r['roofline'])[:900])
PY
  done
done
prec=bf16x3
rm -rf $OUT/prof_train_$prec
(cd /tmp && SEPR_GB_WPS=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$prec -o train -- python $OUT/../bench.py --mode train --steps 2 --warmup 1 --batch 16 --precision $prec > $OUT/prof_train_$prec.log 2>&1)
f=$(find $OUT/prof_train_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_${prec}_wps3.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print('   %-78s %7.1f %8.2f ms %8.1f us' % (r['Name'][:78], int(r['Calls'])/5, float(r['TotalDurationNs'])/5e6, float(r['AverageNs'])/1e3))
PY
find $OUT/prof_train_$prec -name "*kernel_trace.csv" -size +20M -delete
