#!/bin/bash
# rocprofv3 kernel stats of the default inference bench (single pipeline, like the roofline region) -> gpurun_out/r6_kernel_stats.csv
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf /tmp/pb
(cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $OUT/../bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off > /tmp/pb.log 2>&1)
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r6_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sep = [r for r in rows if "sepr::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in sep)
n = 9.0   # forwards in the profile: 2 warm-up + 6 timed + the parity-gate forward
print("sepr kernels %.2f ms per forward (%d forwards assumed)" % (tot / 1e6 / n, n))
for r in sep[:24]:
    print("%7.3f ms/fwd %6.1f calls/fwd %8.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6 / n, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"].replace("sepr::", "").replace("void ", "")[:100]))
PY
tail -2 /tmp/pb.log | cut -c1-300
