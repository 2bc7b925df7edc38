#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -k "large_matches" -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head -40 | cut -c1-900
