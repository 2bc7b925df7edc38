#!/bin/bash
# round 3, call 24: LDS bank conflicts - 144-byte plane rows in the projection core / GCFN middle kernel, row-group-fast lane map in the
# contraction's staging.  Micro-benchmarks, then the train and inference lines per variant, then parity of the combined variant.
set -u
export TMPDIR=/tmp
export GEMM_SHAPES="256000,1536,256;256000,256,768;128000,768,128;128000,128,128;8192,8192,4096"
export GEMM_SKIP_F32=1
for v in "" x3p8; do echo "== projection core, variant '$v'"; SEPR_LIB_VARIANT=$v timeout 200 python tools/gemm_bench.py 2>&1 | grep "M=" | cut -c1-60,95-160; done
for v in "" tnmap; do echo "== contraction, variant '$v'"; SEPR_LIB_VARIANT=$v WGRAD_NORM=1 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "M=" | head -7; done
for v in "" lds144 gbp8 "" lds144; do
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --mode train --steps 4 --warmup 2 --batch 16 --precision bf16x3 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('train variant=[$v]: %.1f utt/s %.2f ms/step loss %.3f' % (r['value'], r['ms_per_step'], r['loss']))"
done
for v in "" x3p8 "" x3p8; do
  SEPR_LIB_VARIANT=$v timeout 300 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision --pmc off 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('large variant=[$v]: %.1f utt/s %.2f ms/step parity %.1f dB' % (r['value'], r['ms_per_step'], r['parity_db_vs_golden']))"
done
SEPR_LIB_VARIANT=lds144 timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "wgrad or gcfn_train or train_step_tiny or full_size" -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -5 | cut -c1-500
