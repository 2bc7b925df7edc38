#!/bin/bash
# time of ONE kernel family inside a bench forward, per library variant:  bash tools/prof_kernel.sh <regex> "" v1 v2 ...
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/prof_kernel.txt
RE="$1"; shift
for v in "$@"; do
  if [[ "$v" == *=* ]]; then e="$v"; else e="SEPR_LIB_VARIANT=$v"; fi
  rm -rf /tmp/pv; (cd /tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o pv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision > /tmp/pv.log 2>&1)
  f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
  python - "$f" "$RE" "$v" <<'PY' | tee -a $OUT/prof_kernel.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows if "sepr::" in r["Name"])
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print("[%s] %-50s n=%4s tot=%8.2f ms avg=%8.1f us max=%8.1f us   (all sepr kernels %.1f ms)" % (sys.argv[3], r["Name"].replace("sepr::", "").replace("void ", "")[:50], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, int(r["MaxNs"]) / 1e3, tot / 1e6))
PY
done
