#!/bin/bash
# PMC pass over the projection micro-benchmark (own run, kernel-trace only - see the profiling rules).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o g -- python $OLDPWD/tools/gemm_bench.py > $OUT/pmc1.log 2>&1
echo "rc=$?"
tail -3 $OUT/pmc1.log
ls $OUT/pmc1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc1/*counter_collection.csv")
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'gemm_kernel' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:40], r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in agg.items():
    print(key, {k: round(sum(v)/len(v)) for k, v in d.items()}, 'n', len(next(iter(d.values()))))
PY
