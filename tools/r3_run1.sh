#!/bin/bash
# round-3 first device run: new GCFN kernels first (own processes), then everything else, then the train bench lines
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/train_check.sh > $OUT/train_check.log 2>&1
cut -c1-600 $OUT/train_summary.txt | tail -60
for prec in bf16x3 bf16; do for b in 16; do
  timeout 300 python bench.py --mode train --steps 3 --warmup 2 --batch $b --precision $prec 2>$OUT/train_bench_${prec}_b$b.err | tee $OUT/train_bench_${prec}_b$b.json | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train $prec B=$b: %.1f utt/s %.1f ms/step (host %.1f) loss %.3f gn %.2f tn avg %.3f ms x %d backend %s' % (r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'], r['loss'], r['grad_norm'], r['roofline']['avg_launch_ms'], r['roofline']['launches'], r['collective_backend']))" | tee -a $OUT/train_summary.txt
  tail -3 $OUT/train_bench_${prec}_b$b.err
done; done
timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-400
