#!/usr/bin/env python3
"""Where does a bench step spend its time?  forward alone / metric alone / both, 10 iterations each, one sync at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.criterion import pit_sisnr
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_sources

dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
model = Model.from_config(cfg, init_seed=0).load_synthetic_(0).eval().to(dev)
B = 32
src = torch.from_numpy(synth_sources(B, 32000, seed=1234)).to(dev)
x = src.sum(1).contiguous()
tgt = src.permute(1, 0, 2).contiguous()


def timed(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} {1e3 * (t2 - t0) / n:8.3f} ms/iter (host enqueue {1e3 * (t1 - t0) / n:7.3f} ms)", flush=True)


out = model(x)
est = torch.stack(out[0], 0)
timed("forward", lambda: model(x))
timed("forward (no aux)", lambda: model.separate(x))
timed("stack", lambda: torch.stack(out[0], 0))
timed("metric", lambda: pit_sisnr(est, tgt, mixture=x))


def both():
    o = model(x)
    e = torch.stack(o[0], 0)
    return pit_sisnr(e, tgt, mixture=x)


timed("forward+stack+metric", both)
os.environ["SEPR_OVERLAP"] = "0"
model2 = Model.from_config(cfg, init_seed=0).load_synthetic_(0).eval().to(dev)
timed("forward (SEPR_OVERLAP=0)", lambda: model2(x))


def both2():
    o = model2(x)
    e = torch.stack(o[0], 0)
    return pit_sisnr(e, tgt, mixture=x)


timed("fwd+metric (OVERLAP=0)", both2)
