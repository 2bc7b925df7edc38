import sys, torch
sys.path.insert(0, sys.argv[1])
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(32, 32000, seed=3).cuda()
ref = None
for it in range(40):
    out = m(x)
    if it in (2, 39):
        torch.cuda.synchronize()
        print(it, "allocated MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20)
    if it == 2:
        ref = [a.clone() for a in out[0]] + [b.clone() for st in out[1] for b in st]
torch.cuda.synchronize()
now = list(out[0]) + [b for st in out[1] for b in st]
print("bitwise stable over 38 overlapped forwards:", all(torch.equal(a, b) for a, b in zip(ref, now)))
