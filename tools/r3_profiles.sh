#!/bin/bash
# round-3 evidence in one call: default bench line (as the driver runs it), rocprofv3 kernel stats of the inference bench and of the
# training bench in both packed precisions, training bench lines at batch 16 / 32, graph vs eager.  Everything lands in gpurun_out/r03/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03; rm -rf $OUT; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s lines=$(wc -l < $OUT/bench.json)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_infer -o infer -- python $OUT/../../bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision > $OUT/prof_infer.log 2>&1)
f=$(find $OUT/prof_infer -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for prec in bf16x3 bf16; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$prec -o train -- python $OUT/../../bench.py --mode train --steps 2 --warmup 1 --batch 16 --precision $prec > $OUT/prof_train_$prec.log 2>&1)
  f=$(find $OUT/prof_train_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_$prec.csv
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
for prec in bf16x3 bf16; do for b in 8 16 32; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b --precision $prec 2>/dev/null > $OUT/train_bench_${prec}_b$b.json
  python -c "
import json; r = json.loads(open('$OUT/train_bench_${prec}_b$b.json').read().strip().split(chr(10))[-1]); print('train $prec B=$b graphs: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f tn %.1f TF' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['roofline']['achieved']))"
done; done
timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch 16 --no-train-graphs 2>/dev/null > $OUT/train_bench_bf16x3_b16_eager.json
python -c "
import json; r = json.loads(open('$OUT/train_bench_bf16x3_b16_eager.json').read().strip().split(chr(10))[-1]); print('train bf16x3 B=16 EAGER: %.1f utt/s %.1f ms/step (host %.1f)' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step']))"
timeout 600 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 5 --warmup 2 > $OUT/large_bench.json 2>/dev/null
python -c "
import json; r = json.loads(open('$OUT/large_bench.json').read().strip().split(chr(10))[-1]); print('large: %.1f utt/s parity %s pit %s cpu %s' % (r['value'], r['parity_db_vs_golden'], r['pit_si_snr_max_abs_delta_db'], (r.get('cpu_baseline') or {}).get('value')))"
python - <<PY
import json
r = json.loads(open("$OUT/bench.json").read().strip().split("\n")[-1])
print("infer: %.1f utt/s %.2f ms/step parity %.1f dB pit %.1e backend %s roof frac %.4f" % (r["value"], r["ms_per_step"], r["parity_db_vs_golden"], r["pit_si_snr_max_abs_delta_db"], r["collective_backend"], r["roofline"]["frac"]))
l = r.get("large", {}); print("large:", {k: l.get(k) for k in ("value", "ms_per_step", "parity_db_vs_golden", "pit_si_snr_max_abs_delta_db", "error")})
for k, t in r.get("train", {}).items(): print("train", k, {q: t.get(q) for q in ("value", "ms_per_step", "host_enqueue_ms_per_step", "loss", "collective_backend", "error")})
print("sub_records_s", r.get("sub_records_s"), "cpu", (r.get("cpu_baseline") or {}).get("value"))
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_criterion.py tests/test_infer.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 | cut -c1-400
