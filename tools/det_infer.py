#!/usr/bin/env python3
"""Run-to-run determinism of the inference forward at bench size (B = 32 x 4 s, Base): how many output elements differ between
repeated calls, per output (main / aux levels), and which utterances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_sources
dev = torch.device("cuda:0")
B = int(os.environ.get("DET_B", "32")); T = int(os.environ.get("DET_T", "32000"))
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).to(dev).eval()
m.pipelines = 1
x = torch.from_numpy(synth_sources(B, T, seed=1234).sum(1)).to(dev)
with torch.no_grad():
    ref_a, ref_x = m(x)
    ref = [a.clone() for a in ref_a] + [a.clone() for lvl in ref_x for a in lvl]
    for it in range(int(os.environ.get("DET_REPS", "4"))):
        a, xx = m(x)
        cur = list(a) + [t for lvl in xx for t in lvl]
        bad = []
        for i, (r, c) in enumerate(zip(ref, cur)):
            d = (r != c)
            if d.any():
                utt = d.reshape(d.shape[0], -1).any(1).nonzero().flatten().tolist() if d.dim() > 1 else []
                bad.append((i, int(d.sum()), float((r - c).abs().max()), utt[:8]))
        print("rep", it, "differing outputs:", bad if bad else "none", flush=True)
