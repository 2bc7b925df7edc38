#!/bin/bash
# batch-1 (4 s) forward as a timeline: rocprofv3 kernel trace of synchronised single forwards -> per forward: first start to last end, busy time and
# gap time per HIP stream (queue), the largest gaps with the kernels around them.  What bounds the single-utterance latency once the kernels are short.
export TMPDIR=/tmp
cat > /tmp/b1t.py <<'PY'
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0, precision="bf16x3").load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(1, 32000, seed=1).cuda()
for _ in range(6):
    m(x)
torch.cuda.synchronize()
time.sleep(0.05)
for _ in range(8):
    m(x)
    torch.cuda.synchronize()
    time.sleep(0.02)          # separates the forwards in the trace
PY
rm -rf /tmp/pb1t; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pb1t -o b1t -- python /tmp/b1t.py $OLDPWD > /tmp/b1t.log 2>&1
f=$(find /tmp/pb1t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sepr::" in r["Kernel_Name"]]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# split into forwards at idle periods > 10 ms
groups, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 10_000_000:
        groups.append(cur); cur = []
    cur.append(e)
groups.append(cur)
groups = [g for g in groups if 150 < len(g) < 400][-8:]
def short(n): return n.replace("sepr::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:44]
for gi, g in enumerate(groups):
    t0, t1 = g[0][0], max(e[1] for e in g)
    line = "forward %d: %d kernels, first start -> last end %.3f ms" % (gi, len(g), (t1 - t0) / 1e6)
    for q in sorted(set(e[2] for e in g)):
        k = [e for e in g if e[2] == q]
        busy = sum(e[1] - e[0] for e in k)
        line += " | queue %s: %d kernels, busy %.3f ms" % (q, len(k), busy / 1e6)
    print(line)
g = groups[-1]
main_q = max(set(e[2] for e in g), key=lambda q: sum(1 for e in g if e[2] == q))
k = [e for e in g if e[2] == main_q]
gaps = sorted(((k[i + 1][0] - k[i][1], short(k[i][3]), short(k[i + 1][3])) for i in range(len(k) - 1)), reverse=True)
tot_gap = sum(x[0] for x in gaps)
print("main queue of the last forward: span %.3f ms, busy %.3f ms, gaps %.3f ms over %d boundaries (median %.2f us)" % (
    (k[-1][1] - k[0][0]) / 1e6, sum(e[1] - e[0] for e in k) / 1e6, tot_gap / 1e6, len(gaps), gaps[len(gaps) // 2][0] / 1e3))
for gp in gaps[:12]:
    print("   gap %7.2f us   after %-44s before %s" % (gp[0] / 1e3, gp[1], gp[2]))
PY
