#!/bin/bash
# review item 4: cost of the fused GCFN epilogue's second read of x - per-kernel time (rocprofv3) and bench rate per library variant
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/r6_resread.txt
for v in "" gfa128 gfa256 "" gfa128 gfa256; do
  rm -rf /tmp/pr
  (cd /tmp && SEPR_LIB_VARIANT=$v SEPR_PIPELINES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o r -- python $OUT/../bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off > /tmp/pr.log 2>&1)
  f=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" <<'PY' | tee -a $OUT/r6_resread.txt
import csv, sys, json
for r in csv.DictReader(open(sys.argv[1])):
    if "gcfn_fused3_kernel<128, 2, 4, 0, false, false, 0>" in r["Name"]:
        print("variant [%-6s] gcfn_fused3_kernel<128,2,4,0>: %4s launches, avg %7.1f us" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
