#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "cla or dropout_contract or tiny_matches or frozen_gates or wgrad or general_loader or learns" 2>&1 | tail -6 | cut -c1-900
timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -q -p no:cacheprovider -k "two_rank_gradients" 2>&1 | tail -4 | cut -c1-900
for v in default gfold32 gfold64; do
  if [ $v = default ]; then unset SEPR_LIB_VARIANT; else export SEPR_LIB_VARIANT=$v; fi
  timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --pmc off --steps 10 > $OUT/bench_fold_$v.json 2> $OUT/bench_fold_$v.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_fold_$v.json").read().strip().split("\n")[-1])
    print("$v", r["value"], "utt/s; single", (r.get("single_pipeline") or {}).get("value"), "; GCFN avg launch ms", r["roofline"]["avg_launch_ms"], "parity", r.get("parity_db_vs_golden"))
except Exception as e:
    print("$v unreadable", e)
PY
done
unset SEPR_LIB_VARIANT
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<PY
import json
r = json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("infer", r["value"], r["ms_per_step"], r.get("parity_ok"), "dp8", r.get("dp8_prediction"))
print("large", {k: r["large"].get(k) for k in ("value", "ms_per_step", "parity_ok", "error")})
for k, t in r["train"].items(): print("train", k, {q: t.get(q) for q in ("value", "ms_per_step", "loss", "error")}, "roof", {q: (t.get("roofline") or {}).get(q) for q in ("frac", "traffic", "traffic_over_algorithmic")})
print("train dp8", (r["train"].get("bf16") or {}).get("dp8_prediction"))
print("sub_records_s", r.get("sub_records_s"), "train_pmc_s", r.get("train_pmc_s"), "pmc_s", r.get("pmc_s"))
print(((r["train"].get("bf16") or {}).get("roofline") or {}).get("traffic_source"))
PY
ls $OUT/bench_train_*.stderr 2>/dev/null
