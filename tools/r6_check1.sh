#!/bin/bash
# round 6, first GPU call: inference parity with the folded head + latency ring, then the A/B of both switches
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_infer.py -m gpu -q -x -p no:cacheprovider \
  -k "test_blocks or test_e2e_golden or test_e2e_pit or test_pit_si_snr or test_gcfn or test_full_size or test_ragged or test_no_padding or test_graph_replay or test_batch_pipelines or test_infer or test_separate or test_load" > $OUT/r6_pytest1.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/r6_pytest1.log
bash tools/r6_ab.sh "SEPR_GF_LAT=0 SEPR_FOLD_HEAD=0" "SEPR_GF_LAT=0 SEPR_FOLD_HEAD=1" "SEPR_GF_LAT=2 SEPR_FOLD_HEAD=1" "SEPR_GF_LAT=3 SEPR_FOLD_HEAD=1" "SEPR_GF_LAT=0 SEPR_FOLD_HEAD=0" "SEPR_GF_LAT=3 SEPR_FOLD_HEAD=1"
tail -20 $OUT/r6_lat_err.log
