#!/bin/bash
# per-kernel time of one bench forward under several library variants / env settings (rocprofv3 kernel stats)
#   bash tools/prof_variants.sh "" ablst "SEPR_X3_GRID=4"      (a word with '=' is an env setting)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/prof_variants.txt
for v in "$@"; do
  if [[ "$v" == *=* ]]; then e="$v"; else e="SEPR_LIB_VARIANT=$v"; fi
  echo "===== [$v]" | tee -a $OUT/prof_variants.txt
  rm -rf /tmp/pv; (cd /tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o pv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision > /tmp/pv.log 2>&1)
  f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY' | tee -a $OUT/prof_variants.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "sepr::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in keep)
print("sepr kernels total %.2f ms over 3 forwards" % (tot / 1e6))
for r in keep[:16]:
    print("  %-62s n=%4s tot=%8.2f ms avg=%8.1f us max=%8.1f us" % (r["Name"].replace("sepr::", "").replace("void ", "")[:62], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
PY
done
