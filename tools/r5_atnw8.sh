#!/bin/bash
# round 5: Base inference attention with 8 waves (128 queries) per workgroup - parity under the variant, then whole-model A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_atnw8
[ "${1:-}" = prof ] || SEPR_LIB_VARIANT=atnw8 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocks or e2e_golden or ragged or intermediate_taps or full_size_batch or no_padding" 2>&1 | tail -4 | tee gpurun_out/r05_atnw8/parity.txt
[ "${1:-}" = prof ] || bash tools/ab_model.sh "" atnw8 "" atnw8 2>&1 | tee gpurun_out/r05_atnw8/ab.txt
# kernel-level: rocprofv3 stats of the attention kernel under both libraries ($1 = prof)
if [ "${1:-}" = prof ]; then
  ROOT=$PWD; cd /tmp
  for v in "" atnw8; do
    rm -rf /tmp/p_$v; SEPR_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o s -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precision > /dev/null 2>&1
    echo "variant [${v:-default}]"; grep -h "relattn" $(find /tmp/p_$v -name "*kernel_stats.csv") | cut -c1-60,140-400 
  done 2>&1 | tee $ROOT/gpurun_out/r05_atnw8/stats.txt
fi
