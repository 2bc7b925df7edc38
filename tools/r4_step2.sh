#!/bin/bash
# round-4 device call: attention / optimizer changes - targeted tests, A/B of the optimizer on the bf16 training step, attention kernel
# times (rocprofv3 kernel stats of a short inference and a short training run), contraction plan sweep (tools/wgrad_bench.py)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/r04b
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -x -k "flat_adamw or captured or ega_ or spkattn" --durations=5 2>&1 | tail -12 | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "attn or ega or relattn or model or golden" 2>&1 | tail -5 | cut -c1-600
for o in flat torch; do
  SEPR_BENCH_OPT=$o timeout 300 python bench.py --mode train --batch 16 --steps 4 --warmup 1 --precision bf16 > $OUT/r04b/train_$o.json 2> $OUT/r04b/train_$o.err
  python - <<PY
import json
r = json.loads(open("$OUT/r04b/train_$o.json").read().strip().split("\n")[-1])
print("train bf16 b16 opt=$o", r.get("value"), r.get("ms_per_step"), "loss", r.get("loss"), "gn", r.get("grad_norm"), r.get("config", {}).get("optimizer", "")[:40])
PY
done
prof() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/r04b/prof_$name
  (cd /tmp && SEPR_PIPELINES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04b/prof_$name -o $name -- python $OUT/../bench.py "$@" > $OUT/r04b/prof_$name.log 2>&1)
  f=$(find $OUT/r04b/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r04b/${name}_kernel_stats.csv
  rm -rf $OUT/r04b/prof_$name
  grep -E "relattn|opt_|multi_tensor|gcfn_fused3_kernel<128, 2, 4, 0, false" $OUT/r04b/${name}_kernel_stats.csv | awk -F'","' '{printf "  %-8s calls %6s avg %9.1f us  %s\n", "'$name'", $2, $4/1000, substr($1,2,90)}'
}
prof infer --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off
prof train --mode train --steps 2 --warmup 1 --batch 16 --precision bf16
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision --pmc off > $OUT/r04b/infer.json 2> $OUT/r04b/infer.err
python - <<PY
import json
r = json.loads(open("$OUT/r04b/infer.json").read().strip().split("\n")[-1])
print("infer", r.get("value"), r.get("ms_per_step"), "single", (r.get("single_pipeline") or {}).get("value"), "parity", r.get("parity_db_vs_golden"), r.get("pit_si_snr_max_abs_delta_db"), r.get("parity_ok"), "large", (r.get("large") or {}).get("value"))
PY
for w in 512 384 256; do
  echo "== wgrad_bench plain-bf16 arithmetic, SEPR_TN_WGS=$w"
  SEPR_TN_WGS=$w WGRAD_X3=2 timeout 120 python tools/wgrad_bench.py 2>&1 | grep "M=" | awk '{print $1, $2, $3, $4, $5}' | tr '\n' ';'
  echo
done
