#!/usr/bin/env python3
"""Host-side cost of one training step (cProfile over 3 steps after warm-up): where do the ~80 ms of enqueue time go?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.criterion import PIT_SISNR_mag, PIT_SISNR_time
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_sources

dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = Model.from_config(cfg, init_seed=0).load_synthetic_(0).to(dev)
model.train()
src = torch.from_numpy(synth_sources(B, 32000, seed=1)).to(dev)
x = src.sum(1).contiguous()
tg = [src[:, s].contiguous() for s in range(2)]
sizes = torch.full((B,), 32000)
ct, cm = PIT_SISNR_time(dev, 2, True), PIT_SISNR_mag(dev, 512, 128, "hann", 4, 2, True, False)
params = list(model.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    audio, aux = model(x)
    lm = [cm(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)]
    loss = (0.6 * ct(estims=audio, input_sizes=sizes, target_attr=tg) + 0.4 * sum(lm) / len(lm)) / 2
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"B={B}: host enqueue {1e3 * th / 3:.1f} ms/step, with sync {1e3 * (time.perf_counter() - t0) / 3:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
