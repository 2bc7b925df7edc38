#!/bin/bash
# round-4 device call: selected tests (KSEL_TRAIN / KSEL_PARITY = pytest -k expressions, empty = skip), then optional bench lines.
#   gpurun -- 'KSEL_TRAIN="full_size" BENCH=default bash tools/r4_check.sh'
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
if [ -n "${KSEL_TRAIN:-}" ]; then
  timeout ${T_TRAIN:-1500} python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "$KSEL_TRAIN" --durations=8 2>&1 | tail -${TAIL:-25} | cut -c1-900
fi
if [ -n "${KSEL_PARITY:-}" ]; then
  timeout ${T_PARITY:-900} python -m pytest tests/test_gpu_parity.py tests/test_criterion.py tests/test_infer.py -m gpu -q -p no:cacheprovider -k "$KSEL_PARITY" 2>&1 | tail -8 | cut -c1-600
fi
for b in ${BENCH:-}; do
  case $b in
    default) timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?";;
    infer)   timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --pmc off > $OUT/bench_infer.json 2> $OUT/bench_infer.err; echo "infer bench rc=$?";;
    large)   timeout 300 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --pmc off > $OUT/bench_large.json 2> $OUT/bench_large.err; echo "large bench rc=$?";;
    train16) timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16 > $OUT/bench_train_bf16_b16.json 2> $OUT/bench_train_bf16_b16.err; echo "train bf16 b16 rc=$?";;
    train16x3) timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16x3 > $OUT/bench_train_bf16x3_b16.json 2> $OUT/bench_train_bf16x3_b16.err; echo "train bf16x3 b16 rc=$?";;
  esac
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), r.get("value"), r.get("unit"), r.get("ms_per_step"), "ms | parity", r.get("parity_db_vs_golden"), r.get("pit_si_snr_max_abs_delta_db"), r.get("parity_ok"),
          "| roof", (r.get("roofline") or {}).get("frac"), (r.get("roofline") or {}).get("avg_launch_ms"), "| single", (r.get("single_pipeline") or {}).get("value"))
    if "large" in r: print("   large:", {k: r["large"].get(k) for k in ("value", "ms_per_step", "parity_db_vs_golden", "parity_ok", "error")})
    for k, t in (r.get("train") or {}).items(): print("   train", k, {q: t.get(q) for q in ("value", "ms_per_step", "loss", "grad_norm", "model_frac_algorithmic", "error")})
    if "alt_precision" in r: print("   alt:", r["alt_precision"])
    if "cpu_baseline" in r: print("   cpu:", r["cpu_baseline"].get("value"), r["cpu_baseline"].get("cores"))
PY
for f in $OUT/bench_*.err; do [ -s "$f" ] && { echo "== $f"; tail -3 "$f" | cut -c1-300; }; done
