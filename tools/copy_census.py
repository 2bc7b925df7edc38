#!/usr/bin/env python3
"""Which Python lines of one eager training step issue device-to-device copies / fills (hipMemcpyAsync / hipMemsetAsync -> __amd_rocclr_copyBuffer /
fillBufferAligned in a kernel trace)?  torch.profiler with stacks over ONE step of bench.py's training loop at batch 4."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def step():
    opt.zero_grad(set_to_none=True)
    audio, aux = m(x)
    loss = (0.6 * ct(estims=audio, input_sizes=sizes, target_attr=tg) + 0.4 * sum(cm(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)) / len(aux)) / 2
    loss.backward()
    opt.step(max_norm=5.0)
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::stack", "aten::_foreach_add_", "aten::zeros", "aten::zeros_like"):
        st = [s for s in (e.stack or []) if "sepreformer_amd" in s or "bench" in s or "tools/" in s]
        cnt[(e.name, st[0].split("/root/")[-1][:110] if st else "<torch internal>")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d}  {name:18s} {where}")
