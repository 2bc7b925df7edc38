#!/bin/bash
# Training-path check on the GPU box: every train test group in its own process (a device fault in one does not hide the others).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/train_check.sh'
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT; rm -f $OUT/train_parity_report.json
: > $OUT/train_summary.txt
for grp in test_train_step_large test_wgrad_core test_wgrad_norm test_gcfn_train test_cla_train test_ega_train test_spkattn_train test_downconv_split_fuse test_front_and_heads test_criteria_backward test_train_step_tiny test_train_step_base test_train_step_full_size test_dropout_contract test_training_loop test_train_graph test_captured_whole_step; do
  timeout 420 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "$grp" -p no:cacheprovider > "$OUT/train_$grp.log" 2>&1
  echo "pytest [$grp] rc=$?" | tee -a $OUT/train_summary.txt
  grep -E "passed|failed|error" "$OUT/train_$grp.log" | tail -2 | tee -a $OUT/train_summary.txt
  grep -E "^E  " "$OUT/train_$grp.log" | head -12 | cut -c1-1500 | tee -a $OUT/train_summary.txt
done
if [ -n "${EXTRA:-}" ]; then bash -c "$EXTRA" 2>&1 | tail -40 | tee -a $OUT/train_summary.txt; fi
