#!/bin/bash
# round-4 evidence: rocprofv3 kernel stats of the inference bench (single pipeline, like the roofline region), of the Large bench and of the
# training bench in bf16; WHAT=infer,large,train selects.  Everything lands in gpurun_out/r04/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04; mkdir -p $OUT
WHAT=${WHAT:-infer,large,train}
prof() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  (cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python $OUT/../../bench.py "$@" > $OUT/prof_$name.log 2>&1)
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${name}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== $name: %d kernels, %.1f ms total" % (len(rows), tot / 1e6))
for r in rows[:14]:
    print("  %5.1f %%  %7d calls  %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
}
case ,$WHAT, in *,infer,*) prof infer --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off;; esac
case ,$WHAT, in *,large,*) prof large --variant SepReformer_Large_DM_WHAMR --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --pmc off;; esac
case ,$WHAT, in *,train,*) prof train_bf16 --mode train --steps 2 --warmup 1 --batch 16 --precision bf16;; esac
