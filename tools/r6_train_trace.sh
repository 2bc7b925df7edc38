#!/bin/bash
# one bf16 training step (batch 16) as an ordered launch list: kernel, grid, duration - gpurun_out/r6_train_trace.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf /tmp/pt
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o t -- python $OUT/../bench.py --mode train --batch ${BATCH:-16} --steps 3 --warmup 1 --precision ${PRECISION:-bf16} > /tmp/pt.log 2>&1)
tail -2 /tmp/pt.log
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/r6_train_trace.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("sepr::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]
# the last step = the launches after the last enc_bwd_pre_kernel-terminated step boundary: take the last len/steps chunk by encoder_kernel markers
idx = [i for i, r in enumerate(rows) if "gcfn_bwd_mid_kernel" in r["Kernel_Name"]]
n_mid = len(idx)
per = 56
last = rows[idx[-per] - 40:] if n_mid >= per else rows
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f us  grid %6s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), short(r["Kernel_Name"])))
PY
wc -l $OUT/r6_train_trace.txt
