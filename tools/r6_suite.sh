#!/bin/bash
# the whole -m gpu suite in ONE process, as the driver runs it, with the slowest tests listed (suite-time budget: <= 540 s)
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=40 ) > $OUT/r6_suite.log 2>&1
echo "rc=$?"; grep -E "passed|failed|error" $OUT/r6_suite.log | tail -3; grep -E "^real" $OUT/r6_suite.log
