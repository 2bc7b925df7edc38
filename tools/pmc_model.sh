#!/bin/bash
# PMC pass over one bench step (own run, kernel-trace only).  $1 = kernel-name regex, rest = counters
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; rm -rf $OUT/pmc2
REGEX="$1"; shift
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$REGEX" --output-format csv -d $OUT/pmc2 -o m -- \
  python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > $OUT/pmc2.log 2>&1
echo "rc=$?"; tail -2 $OUT/pmc2.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc2/*counter_collection.csv")
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    key = r['Kernel_Name'][:60]
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
for key, d in agg.items():
    print(key, {k: round(v) for k, v in d.items()})
PY
