#!/bin/bash
# per-launch-size time of the fused GCFN kernels inside a bench forward (rocprofv3 kernel trace), for each environment setting given
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for e in "$@"; do
  rm -rf /tmp/ps
  (cd /tmp && env $e SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o s -- python $OUT/../bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --pmc off > /tmp/ps.log 2>&1)
  f=$(find /tmp/ps -name "*kernel_trace.csv" | head -1)
  python - "$f" "$e" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gcfn_fused3_kernel" in r["Kernel_Name"] and ", 0, false, false" in r["Kernel_Name"]]
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows), reverse=True)
print("==", sys.argv[2], " (%d launches, total %.1f us)" % (len(d), sum(d)))
groups, cur = [], [d[0]]
for v in d[1:]:                      # launches of one size cluster within ~15 %
    if v > cur[0] * 0.82:
        cur.append(v)
    else:
        groups.append(cur); cur = [v]
groups.append(cur)
for g in groups:
    print("  n=%3d  median %7.1f us  min %7.1f  max %7.1f  total %8.1f" % (len(g), sorted(g)[len(g) // 2], min(g), max(g), sum(g)))
PY
done
