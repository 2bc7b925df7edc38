#!/bin/bash
# exactly what the driver runs at round end: the whole -m gpu suite in ONE process, smoke(), the default bench line
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
t0=$(date +%s)
timeout 2400 python -m pytest tests/ ${PYTEST_X:--x} -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu_all.log 2>&1
echo "pytest -m gpu rc=$? wall=$(( $(date +%s) - t0 )) s"; grep -E "passed|failed|error" $OUT/pytest_gpu_all.log | tail -3 | cut -c1-400
t0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-300
echo "smoke wall=$(( $(date +%s) - t0 )) s"
