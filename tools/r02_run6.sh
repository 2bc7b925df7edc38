#!/bin/bash
# round-2 GPU call 6: the 8-wave / two-group GCFN kernel (v5): parity first, then A/B against v3, then the dropout-fusion
# training checks and the 2-rank shared-GPU bench (deadlock fix)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/run6_summary.txt; : > $S
echo "== v5 GCFN parity (block tests, forced big kernel, e2e goldens, batch-vs-alone)" | tee -a $S
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gcfn or blocks or e2e_golden or bit_identical or batch" 2>&1 | tail -5 | tee -a $S
for k in 3 5; do
  for i in 1 2; do
    SEPR_GF_KERNEL=$k timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('  GF_KERNEL=$k: %.1f utt/s  %.2f ms/step  parity %.1f dB  pit %.1e  gcfn %.1f TF algo (%.3f ms avg x %d) pipe %.3f' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r.get('pit_si_snr_max_abs_delta_db', -1), r['roofline']['achieved'], r['roofline']['avg_launch_ms'], r['roofline']['launches'], r['roofline'].get('mfma_pipe_frac', 0)))" | tee -a $S
  done
done
echo "== full GPU parity suite" | tee -a $S
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $S
echo "== 2 ranks on one GPU (self-launch)" | tee -a $S
timeout 240 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision 2>&1 | tail -1 | cut -c1-600 | tee -a $S
for b in 8; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b 2>/dev/null | tee $OUT/train_bench_b$b.json | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train B=$b: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm']))" | tee -a $S
done
