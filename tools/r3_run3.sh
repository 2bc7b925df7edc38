#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== fused B=4"; timeout 300 python tools/det_check.py 4 0.0 2>&1 | grep -v amdgpu.ids | tail -22
echo "== unfused B=4"; SEPR_TRAIN_FUSE_GCFN=0 timeout 300 python tools/det_check.py 4 0.0 2>&1 | grep -v amdgpu.ids | tail -22
echo "== fused B=1"; timeout 300 python tools/det_check.py 1 0.0 2>&1 | grep -v amdgpu.ids | tail -8
