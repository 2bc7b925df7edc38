#!/bin/bash
# A/B of the wide (128 x 256) projection core: parity, micro-benchmark on the Large / training shapes, Large bench, training bench
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "wide or linear_core_bf16x3" 2>&1 | tail -4 | cut -c1-600
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "(gcfn_train and bf16) or (gcfn_train_full_size and bf16) or tiny_bf16" 2>&1 | tail -4 | cut -c1-600
export GEMM_SKIP_F32=1 GEMM_SHAPES="256000,1536,256;8192,8192,4096;512000,768,128;256000,768,256;256000,256,768;256000,512,256;64000,1536,256;512000,256,128"
for w in 0 1; do echo "== SEPR_X3_WIDE=$w"; SEPR_X3_WIDE=$w timeout 300 python tools/gemm_bench.py 2>&1 | grep -v "^variant" | sed 's/f32 *nan ms *nan TF *nan GB.s |//'; done | tee $OUT/r4_wide_gemm.txt
for w in 0 1; do
  SEPR_X3_WIDE=$w timeout 300 python bench.py --variant SepReformer_Large_DM_WHAMR --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --pmc off > $OUT/bench_large_w$w.json 2> $OUT/bench_large_w$w.err
  SEPR_X3_WIDE=$w timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16 > $OUT/bench_train_bf16_w$w.json 2> $OUT/bench_train_bf16_w$w.err
done
timeout 300 python bench.py --mode train --batch 16 --steps 3 --warmup 1 --precision bf16x3 > $OUT/bench_train_bf16x3.json 2> $OUT/bench_train_bf16x3.err; echo "train bf16x3 rc=$?"
grep -E "Error|error|terminate|what\(\)" $OUT/bench_train_bf16x3.err | head -8 | cut -c1-400
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_large_w*.json") + glob.glob("$OUT/bench_train_bf16*.json")):
    try:
        r = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), r.get("value"), r.get("ms_per_step"), "parity", r.get("parity_db_vs_golden"), r.get("pit_si_snr_max_abs_delta_db"), "loss", r.get("loss"), r.get("grad_norm"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
