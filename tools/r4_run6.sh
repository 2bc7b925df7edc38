set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "plane_staged or (gcfn_train and bf16) or (gcfn_train_full_size and bf16) or tiny_bf16 or base_bf16_matches or learns_bf16 or gcfn_train_unfused" 2>&1 | tail -6 | cut -c1-1200
for pl in 0 1; do
  SEPR_TRAIN_GCFN_PLANES=$pl timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16 > $OUT/bench_train_bf16_pl$pl.json 2> $OUT/bench_train_bf16_pl$pl.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_train_bf16_pl$pl.json").read().strip().split("\n")[-1])
    print("planes=$pl", r["value"], "utt/s", r["ms_per_step"], "ms loss", r["loss"], "gn", r["grad_norm"], "tn frac", r["roofline"]["frac"], r["roofline"]["avg_launch_ms"])
except Exception as e:
    print("planes=$pl unreadable", e)
PY
  tail -2 $OUT/bench_train_bf16_pl$pl.err | cut -c1-300
done
