#!/bin/bash
# A/B of library variants / environment switches on the bf16 training step inside ONE device call (box-to-box variance is +-4 %):
#   gpurun -- 'AB="band16: SEPR_LIB_VARIANT=band16|base:" bash tools/r4_ab_train.sh'      (entries: name: ENV=... ENV=...)
# prints utt/s of bench.py --mode train --batch 16 --precision bf16 and, with PROF=regex, the rocprofv3 average of the matching kernels
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ab; mkdir -p $OUT
IFS='|' read -ra ENTRIES <<< "${AB:-base:}"
for rep in 1 ${REPS:-}; do
for e in "${ENTRIES[@]}"; do
  name=${e%%:*}; envs=${e#*:}
  env $envs timeout 300 python bench.py --mode train --batch ${BATCH:-16} --steps ${STEPS:-4} --warmup 1 --precision ${PRECISION:-bf16} > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/$name.json").read().strip().split("\n")[-1])
    print("%-10s rep $rep  %8.2f utt/s  %7.2f ms  loss %s" % ("$name", r["value"], r["ms_per_step"], r.get("loss")))
except Exception as ex:
    print("$name failed:", ex); print(open("$OUT/$name.err").read()[-800:])
PY
done; done
if [ -n "${PROF:-}" ]; then
for e in "${ENTRIES[@]}"; do
  name=${e%%:*}; envs=${e#*:}
  rm -rf $OUT/prof_$name
  (cd /tmp && env $envs SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python $OUT/../../bench.py --mode train --steps 2 --warmup 1 --batch ${BATCH:-16} --precision ${PRECISION:-bf16} > $OUT/prof_$name.log 2>&1)
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
  python - <<PY
import csv, re
for r in csv.DictReader(open("$OUT/${name}_kernel_stats.csv")):
    if re.search(r"$PROF", r["Name"]):
        print("  %-10s %6d calls %9.1f us  %s" % ("$name", int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
done
fi
