import sys, time, torch
sys.path.insert(0, sys.argv[1])
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
for name in ("SepReformer_Large_DM_WHAMR", "SepReformer_Large_DM_WHAM"):
    m = Model.from_config(VARIANTS[name], init_seed=0).load_synthetic_(0).eval().to("cuda:0")
    for B in (1, 8):
        x = synth_mixture(B, 32000, seed=3).cuda()
        out = m(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            out = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        fin = all(torch.isfinite(a).all().item() for a in out[0])
        print(f"{name} B={B}: {1e3*dt:.1f} ms/step = {B/dt:.1f} utt/s finite={fin} out={tuple(out[0][0].shape)} aux={len(out[1])}")
    del m
    torch.cuda.empty_cache()
