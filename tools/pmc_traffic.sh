#!/bin/bash
# HBM traffic of the dominant kernel per the guide's recipe: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (kernel-trace only), summed over the launches of one bench step and divided by the launch count.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
RE=${1:-gcfn_fused}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$RE" --output-format csv -d /tmp/pmc_$c -o t -- \
     python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > /tmp/pmc_$c.log 2>&1)
done
python3 - <<'PY' | tee $OUT/pmc_traffic.txt
import csv, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    res[c] = (sum(float(r["Counter_Value"]) for r in rows), len(rows))
    print(c, "sum", res[c][0], "over", res[c][1], "launches")
print(json.dumps({k: {"sum": v[0], "launches": v[1]} for k, v in res.items()}))
PY
