#!/bin/bash
# the copyBuffer launches INSIDE a replayed (hipGraph-captured) training step: which kernels do they sit between?
export TMPDIR=/tmp
cat > /tmp/cc3.py <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.criterion import PIT_SISNR_mag, PIT_SISNR_time
from sepreformer_amd.model import Model
from sepreformer_amd.optim import FlatAdamW
from sepreformer_amd.synth import synth_sources
from sepreformer_amd.train_step import CapturedTrainStep
dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
m = Model.from_config(cfg, init_seed=0, precision="bf16").load_synthetic_(0).to(dev).train()
B, T = 4, 32000
src = torch.from_numpy(synth_sources(B, T, seed=1)).to(dev)
x = src.sum(1).contiguous(); tg = [src[:, s].contiguous() for s in range(2)]; sizes = torch.full((B,), T)
ct, cm = PIT_SISNR_time(dev, 2, True), PIT_SISNR_mag(dev, 512, 128, "hann", cfg.num_stages, 2, True, False)
opt = FlatAdamW(m, lr=1e-4, weight_decay=1e-2)
def loss_fn(audio, aux, *tg):
    tg = list(tg)
    return (0.6 * ct(estims=audio, input_sizes=sizes, target_attr=tg) + 0.4 * sum(cm(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)) / len(aux)) / 2
cap = CapturedTrainStep(m, loss_fn, opt, x, tg, max_norm=5.0, warmup=1)
torch.cuda.synchronize()
mark = lambda: torch.arange(7, device=dev).cumsum(0)
mark(); cap(x, tg); mark(); torch.cuda.synchronize()
PY
rm -rf /tmp/pcc3; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pcc3 -o cc -- python /tmp/cc3.py $OLDPWD > /tmp/cc3.log 2>&1
f=$(find /tmp/pcc3 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "cumsum" in n.lower() or "scan" in n.lower()]
a, b = marks[-2], marks[-1]
seg = names[a + 1:b]
c = collections.Counter("copyBuffer" if "copyBuffer" in n else "fillBuffer" if "fillBuffer" in n else "aten" if "at::native" in n else "sepr" for n in seg)
print("one replayed step:", len(seg), "launches", dict(c))
prev, nxt = collections.Counter(), collections.Counter()
short = lambda n: n.replace("void ", "").replace("sepr::", "").replace("(anonymous namespace)::", "")[:60]
for i, n in enumerate(seg):
    if "copyBuffer" in n:
        j = i - 1
        while j >= 0 and "copyBuffer" in seg[j]: j -= 1
        k = i + 1
        while k < len(seg) and "copyBuffer" in seg[k]: k += 1
        prev[short(seg[j]) if j >= 0 else "-"] += 1
        nxt[short(seg[k]) if k < len(seg) else "-"] += 1
print("before copies:", prev.most_common(8))
print("after copies :", nxt.most_common(8))
runs = collections.Counter(); i = 0
while i < len(seg):
    if "copyBuffer" in seg[i]:
        j = i
        while j < len(seg) and "copyBuffer" in seg[j]: j += 1
        runs[j - i] += 1; i = j
    else: i += 1
print("lengths of consecutive copyBuffer runs:", sorted(runs.items()))
PY
