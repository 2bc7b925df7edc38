#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/r04b
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "ega or flat_adamw or captured_step_with or train_step_base_bf16 or learns_bf16 or tiny_bf16" --tb=short 2>&1 | tail -30 | cut -c1-300
for one in 1 0; do
  SEPR_TRAIN_ATTN_ONE=$one timeout 300 python bench.py --mode train --batch 16 --steps 4 --warmup 1 --precision bf16 > $OUT/r04b/train_one$one.json 2> $OUT/r04b/train_one$one.err
  python - <<PY
import json
r = json.loads(open("$OUT/r04b/train_one$one.json").read().strip().split("\n")[-1])
print("train bf16 b16 ATTN_ONE=$one", r.get("value"), r.get("ms_per_step"), "loss", r.get("loss"), "gn", r.get("grad_norm"))
PY
done
prof() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/r04b/prof_$name
  (cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04b/prof_$name -o $name -- python $OUT/../bench.py "$@" > $OUT/r04b/prof_$name.log 2>&1)
  f=$(find $OUT/r04b/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r04b/${name}_kernel_stats.csv
  rm -rf $OUT/r04b/prof_$name
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/r04b/${name}_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("relattn", "gcfn_bwd_mid")):
        print("  %-6s %6d calls %9.1f us  %s" % ("$name", int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
}
prof train3 --mode train --steps 2 --warmup 1 --batch 16 --precision bf16
