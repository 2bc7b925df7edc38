#!/bin/bash
# round 6, last session: the wide projection core after SEPR_XW_REDERIVE = 1 - every test that names the wide core / Large variants, then the Large and bf16 training bench lines
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
( time timeout 280 python -m pytest tests/ -q -x -m gpu -p no:cacheprovider -k "wide or Large or large or WHAM" ) 2>&1 | tail -5
AB="SEPR_X=0" bash tools/r6_large_ab.sh
AB="SEPR_X=0" bash tools/r6_train_ab.sh | head -1
