#!/usr/bin/env python3
"""Round 6 A/B driver (run once per environment setting, e.g. SEPR_GF_LAT=0/2/3, SEPR_FOLD_HEAD=0/1): Base bf16x3,
  * batch-1 latency of model(x) on a 4 s utterance and on the sample_WSJ length (73 593 samples), median of REPS eager forwards (+ hipGraph replay);
  * batch-32 ms per forward (public call, default pipelines), median of 5 x 4 forwards;
  * agreement with the committed golden e2e_base_4s (main outputs) and bitwise equality batch-1-alone vs inside the batch of 32.
Prints ONE line of JSON.  tools/r6_ab.sh runs it over a list of settings inside one gpurun call (box-to-box variance is +-4 %)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sepreformer_oracle as orc                     # noqa: E402  (checker only)
from sepreformer_amd.config import VARIANTS                      # noqa: E402
from sepreformer_amd.model import Model                          # noqa: E402
from sepreformer_amd.synth import synth_sources                  # noqa: E402

REPS = int(os.environ.get("R6_REPS", "25"))
dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
model = Model.from_config(cfg, init_seed=0, precision="bf16x3").load_synthetic_(0).eval().to(dev)
src = torch.from_numpy(synth_sources(32, 32000, seed=1234)).to(dev)
x = src.sum(1).contiguous()


def med_ms(fn, reps=REPS):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t)
    return round(1e3 * sorted(ts)[len(ts) // 2], 3)


rec = {"env": {k: v for k, v in os.environ.items() if k.startswith("SEPR_")}}
x1 = x[:1].contiguous()
out1 = model(x1)
rec["lat_b1_4s_ms"] = med_ms(lambda: model(x1))
xs = torch.from_numpy(synth_sources(1, 73593, seed=7)).to(dev).sum(1).contiguous()
model(xs)
rec["lat_b1_9s_ms"] = med_ms(lambda: model(xs))
model.use_graphs = True
try:
    model(x1)
    rec["lat_b1_4s_graph_ms"] = med_ms(lambda: model(x1))
finally:
    model.use_graphs = False
out32 = model(x)
rec["b32_ms"] = med_ms(lambda: [model(x) for _ in range(4)], reps=5) / 4
rec["b32_utt_s"] = round(32e3 / rec["b32_ms"], 1)
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_base_4s.npz"))
xg = torch.from_numpy(g["x"]).to(dev)
og = model(xg)
rec["golden_db"] = round(min(orc.agreement_db(og[0][s].cpu(), torch.from_numpy(g["main"][s])) for s in range(2)), 2)
rec["alone_vs_batch_bitwise"] = bool(all(torch.equal(out1[0][s][0], out32[0][s][0]) for s in range(2)))
print(json.dumps(rec))
