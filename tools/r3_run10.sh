#!/bin/bash
# round 3, call 10: tail-cleanup kernels (tests + bench with kernel stats), PMC traffic of the weight-gradient contraction
set -u
KSEL="cla_train or front_and_heads or downconv_split_fuse or train_step_tiny or spkattn_train" bash tools/r3_quick.sh
bash tools/pmc_train_tn.sh 2>&1 | tail -40
