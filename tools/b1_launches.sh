#!/bin/bash
# every kernel of a batch-1 (4 s) forward with its launches per forward and time: the launch-count budget of the single-utterance path
export TMPDIR=/tmp
cat > /tmp/b1.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(1, 32000, seed=1).cuda()
m(x); torch.cuda.synchronize()
import time
for _ in range(10):
    m(x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    m(x)
torch.cuda.synchronize()
print("eager ms per forward", (time.perf_counter() - t) * 100)
PY
rm -rf /tmp/pb1; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o b1 -- python /tmp/b1.py $OLDPWD 2>&1 | grep "eager ms"
f=$(find /tmp/pb1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "sepr::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in keep); n = sum(int(r["Calls"]) for r in keep)
print("sepr kernels: %.2f ms/forward, %.0f launches/forward" % (tot / 21e6, n / 21))
for r in keep:
    print("  %-70s n/fwd=%6.1f tot/fwd=%7.1f us avg=%7.1f us" % (r["Name"].replace("sepr::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:70], int(r["Calls"]) / 21, int(r["TotalDurationNs"]) / 21e3, float(r["AverageNs"]) / 1e3))
PY
