#!/bin/bash
# parity of a library variant on the block + e2e tests, then an A/B bench:  bash tools/v4check.sh <variant>
export TMPDIR=/tmp
V=${1:-gfv4}
SEPR_LIB_VARIANT=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_blocks or test_e2e_golden or test_full_size" -p no:cacheprovider 2>&1 | tail -3
bash tools/ab_model.sh "" $V "" $V
