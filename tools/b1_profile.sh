#!/bin/bash
# per-kernel time of a batch-1 (4 s) forward: what bounds the single-utterance latency
export TMPDIR=/tmp
cat > /tmp/b1.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(1, 32000, seed=1).cuda()
for _ in range(6):
    m(x)
torch.cuda.synchronize()
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b1 -- python /tmp/b1.py $OLDPWD > /tmp/b1.log 2>&1
f=$(find /tmp/pb1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "sepr::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in keep)
print("sepr kernels total %.2f ms over 6 forwards = %.2f ms/forward" % (tot / 1e6, tot / 6e6))
for r in keep[:14]:
    print("  %-56s n=%4s tot=%7.2f ms avg=%7.1f us" % (r["Name"].replace("sepr::", "").replace("void ", "")[:56], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
