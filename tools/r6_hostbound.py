import sys, time, torch
sys.path.insert(0, "/root/repo")
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model
from sepreformer_amd.synth import synth_mixture
m = Model.from_config(VARIANTS["SepReformer_Base_WSJ0"], init_seed=0, precision="bf16x3").load_synthetic_(0).eval().to("cuda:0")
x = synth_mixture(1, 32000, seed=1).cuda()
for _ in range(5): m(x)
torch.cuda.synchronize()
# (a) enqueue-only time of 20 forwards (no sync inside), then the drain
t0 = time.perf_counter()
for _ in range(20): m(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue per forward %.3f ms, drain after %.3f ms, total per forward %.3f ms" % ((t1 - t0) * 50, (t2 - t1) * 1e3, (t2 - t0) * 50))
# (b) one forward at a time with sync
ts = []
for _ in range(25):
    torch.cuda.synchronize(); t = time.perf_counter(); m(x); th = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t, th - t))
ts.sort()
print("single forward: %.3f ms wall, host part %.3f ms" % (ts[12][0] * 1e3, ts[12][1] * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
