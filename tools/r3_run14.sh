#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== plain"; timeout 200 python tools/wgrad_bench.py 2>&1 | grep "M=" | head -4
echo "== norm"; WGRAD_NORM=1 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "M=" | head -4
echo "== norm, rocprof"; cd /tmp; WGRAD_NORM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg -o w -- python $OLDPWD/tools/wgrad_bench.py 2>&1 | grep "M=" | head -4
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/wg/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'gemm_tn' in r['Kernel_Name']]
import collections
d=collections.defaultdict(list)
for r in rows: d[int(r['Grid_Size_X'])//256].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k,len(v),'avg %.1f min %.1f'%(sum(v)/len(v),min(v)))
PY
