#!/bin/bash
# as rccl_watchdog_loop.sh, but in the situation of the default bench run: a PARENT process with a live HIP context and a 1-rank RCCL group
# (that has run a collective) sits on the same GPU while the training sub-runs start, capture and run
OUT=gpurun_out/r05_watchdog; mkdir -p $OUT
export TMPDIR=/tmp NCCL_DEBUG=WARN TORCH_SHOW_CPP_STACKTRACES=1
N=${N:-30}
python - <<'PY' &
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(3, device="cuda:0", dtype=torch.float64)
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("HOLD_S", "600")))
PY
HOLDER=$!
sleep 8
for mode in ${MODES:-0}; do
  fail=0
  for i in $(seq 1 $N); do
    SEPR_CAPTURE_DRAIN_S=$mode MASTER_PORT=$((29600 + i)) timeout 200 python bench.py --mode train --batch 4 --steps 1 --warmup 1 --precision bf16 > $OUT/h_run_${mode}_$i.json 2> $OUT/h_run_${mode}_$i.err
    rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "holder, drain=$mode run $i rc=$rc"; tail -40 $OUT/h_run_${mode}_$i.err > $OUT/H_FAILED_${mode}_$i.txt; else rm -f $OUT/h_run_${mode}_$i.err $OUT/h_run_${mode}_$i.json; fi
  done
  echo "with a parent process group on the GPU, drain=$mode: $fail failures in $N runs"
done | tee $OUT/summary2.txt
kill $HOLDER 2>/dev/null
