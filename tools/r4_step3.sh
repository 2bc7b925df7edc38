#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/r04b
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "flat_adamw or captured_step_with or ega_train" --tb=short 2>&1 | tail -60 | cut -c1-400
prof() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/r04b/prof_$name
  (cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04b/prof_$name -o $name -- python $OUT/../bench.py "$@" > $OUT/r04b/prof_$name.log 2>&1)
  f=$(find $OUT/r04b/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r04b/${name}_kernel_stats.csv
  rm -rf $OUT/r04b/prof_$name
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/r04b/${name}_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("relattn", "opt_", "gcfn_bwd_mid")):
        print("  %-6s %6d calls %9.1f us  %s" % ("$name", int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
}
prof train2 --mode train --steps 2 --warmup 1 --batch 16 --precision bf16
for w in 512 384 256; do
  echo "== wgrad_bench plain-bf16 arithmetic, SEPR_TN_WGS=$w"
  SEPR_TN_WGS=$w WGRAD_X3=2 timeout 120 python tools/wgrad_bench.py 2>&1 | grep "M=" | cut -c1-52
done
