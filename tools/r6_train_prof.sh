#!/bin/bash
# bf16 training bench (batch 16) + rocprofv3 kernel stats -> gpurun_out/r6_train_*
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python bench.py --mode train --batch 16 --steps 6 --warmup 2 --precision bf16 2>$OUT/r6_train_err.log | tail -1 > $OUT/r6_train_bench.json
python -c "
import json; r = json.loads(open('$OUT/r6_train_bench.json').read()); print('train bf16 b16: %.1f utt/s  %.2f ms/step  loss %s' % (r['value'], r['ms_per_step'], r.get('loss')))"
rm -rf /tmp/pt
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python $OUT/../bench.py --mode train --batch 16 --steps 4 --warmup 1 --precision bf16 > /tmp/pt.log 2>&1)
f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r6_train_kernel_stats_bf16.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("all kernels %.1f ms" % (tot / 1e6))
for r in rows[:32]:
    print("%8.2f ms %6s calls %8.1f us %5s%%  %s" % (int(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5], r["Name"].replace("sepr::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:95]))
PY
