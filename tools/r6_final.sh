#!/bin/bash
# round 6 final evidence: the default bench line, rocprofv3 kernel stats of the inference bench and of the bf16 training step
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=${TAG:-v7}
( time python bench.py > $OUT/r6_bench_$TAG.json 2> $OUT/r6_bench_$TAG.err ) 2>&1 | grep real
python - <<PY
import json
r = json.loads(open("$OUT/r6_bench_$TAG.json").read().strip().splitlines()[-1])
print(r.get("summary"))
PY
bash tools/r6_prof.sh 2>&1 | tee $OUT/r6_kernel_summary.txt | head -12
bash tools/r6_train_prof.sh 2>&1 | tee $OUT/r6_train_prof.txt | head -14
