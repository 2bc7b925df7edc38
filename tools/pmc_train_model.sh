#!/bin/bash
# PMC passes over one eager training step (kernel-trace only, one counter group per run): $1 = kernel-name regex, rest = counters
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; rm -rf $OUT/pmc3
REGEX="$1"; shift
ROOT=$PWD
cd /tmp
timeout 500 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$REGEX" --output-format csv -d $OUT/pmc3 -o m -- \
  python $ROOT/bench.py --mode train --precision ${PMC_PREC:-bf16x3} --batch 16 --steps 1 --warmup 0 --train-graphs off > $OUT/pmc3.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc3/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows:
    key = r['Kernel_Name'][:70]
    agg[key][r['Counter_Name']] += float(r['Counter_Value']); disp[key].add(r['Dispatch_Id'])
for key, d in agg.items():
    print(key, 'launches', len(disp[key]), {k: round(v) for k, v in d.items()})
PY
