#!/bin/bash
# rocprofv3 kernel stats of the Large bench (single pipeline)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf /tmp/plg
(cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/plg -o lg -- python $OUT/../bench.py --variant SepReformer_Large_DM_WHAMR --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --pmc off > /tmp/plg.log 2>&1)
f=$(find /tmp/plg -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r6_large_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total %.1f ms (5 forwards)" % (tot / 1e6))
for r in rows[:22]:
    print("%8.2f ms %5s calls %8.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].replace("sepr::","").replace("void ","")[:100]))
PY
