#!/usr/bin/env python3
"""Which gradient tensors differ between two identical training steps (Base, 4 s)?  usage: det_check.py [B] [p_drop]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.criterion import PIT_SISNR_mag, PIT_SISNR_time
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_sources

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
dev = torch.device("cuda:0")
T = 32000
src = torch.from_numpy(synth_sources(B, T, seed=77)).to(dev)
x = src.sum(1).contiguous()
tg = [src[:, s].contiguous() for s in range(2)]
sizes = torch.full((B,), T)


def run():
    torch.manual_seed(1)
    cfg = dataclasses.replace(VARIANTS["SepReformer_Base_WSJ0"], dropout=p)
    m = Model.from_config(cfg, init_seed=0, precision=os.environ.get("PREC", "bf16x3")).load_synthetic_(0).to(dev)
    m.train()
    audio, aux = m(x)
    ct, cm = PIT_SISNR_time(dev, 2, True), PIT_SISNR_mag(dev, 512, 128, "hann", 4, 2, True, False)
    lm = [cm(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)]
    loss = (0.6 * ct(estims=audio, input_sizes=sizes, target_attr=tg) + 0.4 * sum(lm) / len(lm)) / 2
    loss.backward()
    return {k: v.grad.clone() for k, v in m.named_parameters()}, [a.detach().clone() for a in audio]


g1, a1 = run()
g2, a2 = run()
print("outputs equal:", all(torch.equal(u, v) for u, v in zip(a1, a2)))
bad = [(k, float((g1[k] - g2[k]).abs().max()), float(g1[k].abs().max())) for k in g1 if not torch.equal(g1[k], g2[k])]
print(f"{len(bad)} of {len(g1)} gradient tensors differ")
print("first differing tensors in BACKWARD order (the origin is the first one):")
for k, d, s in bad[::-1][:int(os.environ.get("NSHOW", "14"))]:
    print(f"  {k}: max abs diff {d:.3e} (scale {s:.3e})")
names = list(g1)
last = names.index(bad[-1][0]) if bad else -1
print("identical tensors after the last differing one (backward: before it):", [n for n in names[last + 1:]][:12])
