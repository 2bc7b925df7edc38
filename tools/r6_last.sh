#!/bin/bash
# round 6, last session: the whole -m gpu suite in one process, then the final evidence (bench line + both rocprofv3 kernel-stats passes); AB=1 adds the
# chunks-per-block A/B of the depthwise weight gradient (SEPR_DWWG_NC)
bash tools/r6_suite.sh
TAG=${TAG:-v9} bash tools/r6_final.sh
if [ -n "${AB:-}" ]; then
  AB="SEPR_DWWG_NC=4 SEPR_DWWG_NC=8 SEPR_DWWG_NC=4 SEPR_DWWG_NC=8" bash tools/r6_train_ab.sh | tee gpurun_out/r6_dwwg_ab2.txt
fi
