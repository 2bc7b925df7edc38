#!/bin/bash
# round-4 final evidence: the default bench line as the driver runs it, then rocprofv3 kernel stats (inference, Large, bf16 training)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<PY
import json
r = json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("infer", r["value"], r["ms_per_step"], "single", (r.get("single_pipeline") or {}).get("value"), "parity", r["parity_db_vs_golden"], r["pit_si_snr_max_abs_delta_db"], r.get("parity_ok"))
print("roof", {k: r["roofline"].get(k) for k in ("frac", "avg_launch_ms", "traffic", "traffic_over_algorithmic")}, r["roofline"].get("traffic_source", "")[:80])
print("alt", r.get("alt_precision"), "lat", r.get("latency_b1"))
l = r["large"]; print("large", {k: l.get(k) for k in ("value", "ms_per_step", "parity_db_vs_golden", "parity_ok", "error")}, "roof", {k: (l.get("roofline") or {}).get(k) for k in ("frac", "traffic", "traffic_over_algorithmic")}, "cpu", l.get("cpu_baseline"))
for k, t in r["train"].items(): print("train", k, {q: t.get(q) for q in ("value", "ms_per_step", "loss", "model_frac_algorithmic", "error")}, "roof", {q: (t.get("roofline") or {}).get(q) for q in ("frac", "traffic_over_algorithmic")})
print("cpu", (r.get("cpu_baseline") or {}).get("value"), "sub_records_s", r.get("sub_records_s"), "pmc_s", r.get("pmc_s"), "train_pmc_s", r.get("train_pmc_s"), "gate_failures", r.get("gate_failures"))
PY
ls $OUT/bench_train_*.stderr 2>/dev/null
WHAT=infer,large,train bash tools/r4_profiles.sh 2>&1 | grep -E "^==|gcfn_bwd_mid|relattn|gemm_tn_kernel|gcfn_fused3_kernel<128, 2, 4, 0" | cut -c1-150
