#!/bin/bash
# first contact of the bf16x3 core: core test, blocks, micro-benchmark, model bench in both precisions
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "linear_core" -p no:cacheprovider 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_blocks" -p no:cacheprovider 2>&1 | tail -25
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
for p in fp32 bf16x3; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --precision $p 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[$p]: %.1f utt/s  %.2f ms/step  parity %.1f dB  roofline %s' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], json.dumps(r['roofline'])[:300]))"
done
