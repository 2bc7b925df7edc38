#!/bin/bash
# fused GCFN: parity on blocks / e2e, then A/B bench (SEPR_FUSE_GCFN=0/1)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_blocks or test_e2e_golden or test_intermediate or test_ragged" -p no:cacheprovider 2>&1 | tail -25
for f in 0 1; do
  SEPR_FUSE_GCFN=$f timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench[fuse=$f]: %.1f utt/s  %.2f ms/step  parity %.1f dB  site avg %.3f ms over %d launches' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden'], r['roofline']['avg_launch_ms'], r['roofline']['launches']))"
done
