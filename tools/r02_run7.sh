#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
SEPR_LIB_VARIANT=gf5trace timeout 300 python tools/gf5_trace.py 64 4000 2>&1 | grep -v amdgpu.ids | tee $OUT/gf5_trace.txt | tail -30
echo "== full GPU parity suite"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 | tee $OUT/run7_tests.txt
