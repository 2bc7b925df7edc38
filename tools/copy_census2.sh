#!/bin/bash
# where in an eager training step do the __amd_rocclr_copyBuffer / fillBufferAligned launches come from?  kernel trace with marker kernels between phases
export TMPDIR=/tmp
cat > /tmp/cc2.py <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.criterion import PIT_SISNR_mag, PIT_SISNR_time
from sepreformer_amd.model import Model
from sepreformer_amd.optim import FlatAdamW
from sepreformer_amd.synth import synth_sources
dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
m = Model.from_config(cfg, init_seed=0, precision="bf16").load_synthetic_(0).to(dev).train()
B, T = 4, 32000
src = torch.from_numpy(synth_sources(B, T, seed=1)).to(dev)
x = src.sum(1).contiguous(); tg = [src[:, s].contiguous() for s in range(2)]; sizes = torch.full((B,), T)
ct, cm = PIT_SISNR_time(dev, 2, True), PIT_SISNR_mag(dev, 512, 128, "hann", cfg.num_stages, 2, True, False)
opt = FlatAdamW(m, lr=1e-4, weight_decay=1e-2)
mark = lambda: torch.arange(7, device=dev).cumsum(0)      # a kernel nothing else launches: phase marker
def step(marks):
    opt.zero_grad(set_to_none=True)
    if marks: mark()
    audio, aux = m(x)
    if marks: mark()
    loss = (0.6 * ct(estims=audio, input_sizes=sizes, target_attr=tg) + 0.4 * sum(cm(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)) / len(aux)) / 2
    if marks: mark()
    loss.backward()
    if marks: mark()
    opt.step(max_norm=5.0)
    if marks: mark()
step(False); step(False); torch.cuda.synchronize()
step(True); torch.cuda.synchronize()
PY
rm -rf /tmp/pcc; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pcc -o cc -- python /tmp/cc2.py $OLDPWD > /tmp/cc2.log 2>&1
f=$(find /tmp/pcc -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "cumsum" in n.lower() or "scan" in n.lower()]
print("kernels", len(names), "marker kernels at", marks[-12:])
marks = marks[-5:]
phases = ["forward (incl. weight re-pack)", "criteria", "backward", "optimizer"]
for p, (a, b) in zip(phases, zip(marks, marks[1:])):
    seg = names[a + 1:b]
    c = collections.Counter("copyBuffer" if "copyBuffer" in n else "fillBuffer" if "fillBuffer" in n else "aten" if "at::native" in n else "sepr" for n in seg)
    print(f"{p:32s} {len(seg):5d} launches: {dict(c)}")
    prev = collections.Counter()
    for i, n in enumerate(seg):
        if "copyBuffer" in n and i > 0:
            prev[seg[i - 1][:70]] += 1
    print("    kernels launched right BEFORE a copyBuffer:", prev.most_common(6))
PY
