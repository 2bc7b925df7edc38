set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "plane_staged or (gcfn_train and bf16) or (full_size and bf16) or ega_train or tiny_bf16 or base_bf16_matches or learns_bf16" 2>&1 | tail -5 | cut -c1-1200
for t in 0 1; do
  SEPR_TN16=$t timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16 > $OUT/bench_train_bf16_tn$t.json 2> $OUT/bench_train_bf16_tn$t.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_train_bf16_tn$t.json").read().strip().split("\n")[-1])
    print("tn16=$t", r["value"], "utt/s", r["ms_per_step"], "ms loss", r["loss"], "gn", r["grad_norm"], "tn frac", r["roofline"]["frac"], r["roofline"]["avg_launch_ms"])
except Exception as e:
    print("tn16=$t unreadable", e)
PY
done
timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16x3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('bf16x3', r['value'], r['ms_per_step'], r['loss'])"
