#!/bin/bash
# round-3 validation: full training test groups, bench GPU tests, the DEFAULT bench line (as the driver runs it), inference parity
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/train_check.sh > $OUT/train_check.log 2>&1
cut -c1-500 $OUT/train_summary.txt | tail -44
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-600
t0=$(date +%s)
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s, stdout lines: $(wc -l < $OUT/bench_default.json)"
python - <<PY
import json
r = json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("infer: %.1f utt/s %.2f ms/step parity %.1f dB pit %.1e backend %s roof frac %.4f" % (r["value"], r["ms_per_step"], r["parity_db_vs_golden"], r["pit_si_snr_max_abs_delta_db"], r["collective_backend"], r["roofline"]["frac"]))
print("alt:", r.get("alt_precision")); print("lat:", r.get("latency_b1"))
l = r.get("large", {}); print("large:", {k: l.get(k) for k in ("value", "ms_per_step", "parity_db_vs_golden", "pit_si_snr_max_abs_delta_db", "model_mfma_frac", "error")})
for k, t in r.get("train", {}).items(): print("train", k, {q: t.get(q) for q in ("value", "ms_per_step", "host_enqueue_ms_per_step", "loss", "grad_norm", "collective_backend", "allreduce_bytes_per_step", "model_frac_algorithmic", "error")})
print("sub_records_s", r.get("sub_records_s")); c = r.get("cpu_baseline", {}); print("cpu:", c.get("value"), c.get("cores"), c.get("oracle_over_ref_walltime"))
PY
tail -3 $OUT/bench_default.err | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_criterion.py tests/test_infer.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 | cut -c1-400
