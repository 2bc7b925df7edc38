#!/bin/bash
# Review round 4, fault (c): a training sub-run died in the RCCL watchdog thread (SIGABRT before its first replay) about once in 15 runs.
# N short captured training runs with the capture's drain wait OFF (round-4 behaviour) and ON; stderr of every failed run is kept.
#   N=20 bash tools/rccl_watchdog_loop.sh
OUT=gpurun_out/r05_watchdog; mkdir -p $OUT
export TMPDIR=/tmp NCCL_DEBUG=WARN TORCH_SHOW_CPP_STACKTRACES=1
N=${N:-20}
for mode in 0 0.25; do
  fail=0
  for i in $(seq 1 $N); do
    SEPR_CAPTURE_DRAIN_S=$mode timeout 200 python bench.py --mode train --batch 4 --steps 1 --warmup 1 --precision bf16 > $OUT/run_${mode}_$i.json 2> $OUT/run_${mode}_$i.err
    rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "drain=$mode run $i rc=$rc"; tail -30 $OUT/run_${mode}_$i.err > $OUT/FAILED_${mode}_$i.txt; else rm -f $OUT/run_${mode}_$i.err $OUT/run_${mode}_$i.json; fi
  done
  echo "drain=$mode: $fail failures in $N runs"
done | tee $OUT/summary.txt
