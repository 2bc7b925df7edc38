#!/bin/bash
# round 3, call 22: default bench line with the training sub-records in their own processes (whole-step capture), bench tests
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s, stdout lines: $(wc -l < $OUT/bench_default.json)"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read().strip().split("\n")[-1])
print("infer: %.1f utt/s %.2f ms/step parity %.1f dB" % (r["value"], r["ms_per_step"], r["parity_db_vs_golden"]), "pmc_s", r.get("pmc_s"), "sub_s", r.get("sub_records_s"))
print("traffic_source", r["roofline"]["traffic_source"][:80]); print("single_pipeline", r.get("single_pipeline"), "pipelines", r["config"].get("pipelines"), "roof frac", r["roofline"]["frac"], "avg_launch_ms", r["roofline"]["avg_launch_ms"], "launches", r["roofline"]["launches"]); print("large single", r["large"].get("config",{}).get("pipelines"))
print("large:", r["large"].get("value"), r["large"].get("error"))
for k, t in r.get("train", {}).items(): print("train", k, {q: t.get(q) for q in ("value", "ms_per_step", "host_enqueue_ms_per_step", "host_loop_ms_per_step", "capture_fallback", "collective_backend", "allreduce_bytes_per_step", "error")}); print("   ", (t.get("config") or {}).get("step_launch"))
PY
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -8 | cut -c1-800
