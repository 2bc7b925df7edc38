#!/usr/bin/env python3
"""How long the host needs to enqueue one forward (returns before the GPU finishes) vs the GPU time of the step."""
import sys, time, torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ".")
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(32, 32000, seed=3).cuda()
for p in (1, 2):
    m.pipelines = p
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter(); m(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"pipelines={p}: host enqueue {1e3*sorted(enq)[2]:.1f} ms, step {1e3*sorted(tot)[2]:.1f} ms")
