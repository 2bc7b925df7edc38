#!/bin/bash
# round 3, call 16: default bench line end to end (with the live PMC traffic passes), the live-PMC test
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
t0=$(date +%s.%N)
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
t1=$(date +%s.%N)
echo "default bench wall $(echo "$t1 - $t0" | bc) s, stdout lines $(wc -l < $OUT/bench_default.json)"
python - <<'PY'
import json
r=json.load(open("gpurun_out/bench_default.json"))
print("value", r["value"], "ms", r["ms_per_step"], "parity", r["parity_db_vs_golden"], "pmc_s", r.get("pmc_s"), "sub_s", r.get("sub_records_s"))
print(json.dumps(r["roofline"])[:1500])
print("large", r["large"]["value"], r["large"]["parity_db_vs_golden"])
for k,v in r["train"].items(): print("train", k, v.get("value"), v.get("ms_per_step"), "host", v.get("host_enqueue_ms_per_step"), v.get("error"))
print(json.dumps(r["train"]["bf16x3"]["roofline"])[:700])
PY
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q -x -k "live" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-900
