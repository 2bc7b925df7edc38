export TMPDIR=/tmp
SEPR_LIB_VARIANT=gfv4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_blocks or test_e2e_golden or test_full_size" -p no:cacheprovider 2>&1 | tail -3
bash tools/ab_model.sh "" gfv4
