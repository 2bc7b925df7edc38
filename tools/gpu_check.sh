#!/bin/bash
# One gpurun call: smoke, GPU parity tests (grouped so a device fault in one group does not hide the
# others), bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT; rm -f $OUT/parity_report.json
STEPS=${STEPS:-5}
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)" >> $OUT/device.txt
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | tee -a $OUT/summary.txt
for grp in test_linear_core test_blocks "test_gcfn_block_large or test_gcfn_small_rows" "test_e2e_golden or test_intermediate" "test_e2e_pit or test_pit_si_snr" "test_ragged or test_no_padding or test_error or test_weights or test_groupnorm or test_graph or test_two_replicas or test_dwconv" test_full_size; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$grp" -p no:cacheprovider > "$OUT/pytest_$name.log" 2>&1
  echo "pytest [$grp] rc=$?" | tee -a $OUT/summary.txt
  tail -15 "$OUT/pytest_$name.log" | grep -E "passed|failed|error|Error|dB" | tail -8 | tee -a $OUT/summary.txt
done
timeout 600 python -m pytest tests/test_criterion.py tests/test_infer.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_criterion_infer.log" 2>&1
echo "pytest [criterion + infer] rc=$?" | tee -a $OUT/summary.txt
tail -3 "$OUT/pytest_criterion_infer.log" | grep -E "passed|failed|error" | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps $STEPS --warmup 2 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.log | tee -a $OUT/summary.txt
tail -5 $OUT/bench.err
if [ -n "${AB_ENV:-}" ]; then   # optional A/B leg: AB_ENV="SEPR_LEGACY_POINTWISE=1" etc.
  echo "== bench with $AB_ENV" | tee -a $OUT/summary.txt
  env $AB_ENV timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-precision 2>> $OUT/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], 'utt/s', r['ms_per_step'], 'ms/step parity', r['parity_db_vs_golden'])" | tee -a $OUT/summary.txt
fi
if [ -n "${LARGE:-1}" ]; then   # BASELINE configs[3]: Large_DM_WHAMR (F = 256), B = 32 x 4 s
  echo "== bench Large_DM_WHAMR" | tee -a $OUT/summary.txt
  timeout 600 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision > $OUT/bench_large.log 2>> $OUT/bench.err; echo "bench large rc=$?" | tee -a $OUT/summary.txt
  cat $OUT/bench_large.log | tee -a $OUT/summary.txt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_large -o bench -- python $OLDPWD/bench.py --variant SepReformer_Large_DM_WHAMR --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision --no-metric > $OLDPWD/$OUT/prof_large.log 2>&1)
  f=$(find $OUT/prof_large -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200 | tee -a $OUT/summary.txt
  find $OUT/prof_large -name "*kernel_trace.csv" -size +20M -delete
fi
echo "== rocprofv3 kernel trace" | tee -a $OUT/summary.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision > $OLDPWD/$OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OLDPWD/$OUT/summary.txt
cd $OLDPWD
find $OUT/prof -name "*stats*" | head; 
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200 | tee -a $OUT/summary.txt
# keep the merged-back payload small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
