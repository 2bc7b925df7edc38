#!/usr/bin/env python3
"""Shape robustness: odd batch sizes and lengths through the default path; every utterance must equal its single-utterance
run bit for bit (no kernel may depend on where a row sits in the batch)."""
import sys
import torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ".")
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture

m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
ok = True
for B, T in [(1, 32000), (2, 16000), (3, 9999), (5, 8004), (7, 4001), (33, 8000), (64, 2000), (2, 73596)]:
    x = synth_mixture(B, T, seed=100 + B).cuda()
    out, aux = m(x)
    full = torch.stack(list(out), 0).clone()
    auxf = torch.stack([torch.stack(list(a), 0) for a in aux], 0).clone()
    fin = bool(torch.isfinite(full).all() and torch.isfinite(auxf).all())
    same = True
    for b in sorted({0, B // 2, B - 1}):
        o1, a1 = m(x[b:b + 1])
        same &= torch.equal(torch.stack(list(o1), 0)[:, 0], full[:, b])
        same &= torch.equal(torch.stack([torch.stack(list(a), 0) for a in a1], 0)[:, :, 0], auxf[:, :, b])
    print(f"B={B:3d} T={T:6d} finite={fin} batch-vs-alone-bitwise={same} out={tuple(full.shape)}")
    ok &= fin and same
print("ALL OK" if ok else "FAILURES")
