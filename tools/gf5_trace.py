#!/usr/bin/env python3
"""Block timing / phase timeline of the experimental 8-wave GCFN kernel (csrc/sepr_gcfn_fused5.inc).
  make -C sepreformer_amd/csrc ../_native/libsepr_hip_gf5.so ../_native/libsepr_hip_gf5acc.so
  SEPR_LIB_VARIANT=gf5 SEPR_GF_KERNEL=5 python tools/gf5_trace.py [n T]      # us per launch (SEPR_GF_KERNEL=3: product kernel)
  SEPR_LIB_VARIANT=gf5acc SEPR_GF_KERNEL=5 python tools/gf5_trace.py          # + cycles per phase kind (accumulators)
  SEPR_LIB_VARIANT=gf5trace SEPR_GF_KERNEL=5 python tools/gf5_trace.py        # + s_memtime stamps (perturbs the vmcnt scheme)

Stamp ids: 1 half-step start (U), 2 DMA issued, 3 up-projection issued + seam frames published, 4 after barrier (CD start),
5 DMA issued, 6 conv + GLU + split done, 7 down-projection issued, (barrier), 8 tile change start, 9 tile change done."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sepreformer_amd import lib as L
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.model import Model

n, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4000)
dev = torch.device("cuda:0")
cfg = VARIANTS["SepReformer_Base_WSJ0"]
m = Model.from_config(cfg, init_seed=0).load_synthetic_(0).eval().to(dev)
eng = m.engine()
eng.prepare(max(1, (n * T) // 2400 + 1), 2400, 2400)
x = torch.randn(n, T, cfg.feat, device=dev)
w = eng.pk.enc_stages[0]["g"][0][1]
for _ in range(3):
    y = eng.gcfn(x, w, n, T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    y = eng.gcfn(x, w, n, T)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"[{os.environ.get('SEPR_LIB_VARIANT', 'default')} GF_KERNEL={os.environ.get('SEPR_GF_KERNEL', '5')}] M = {n * T} rows: "
      f"{dt * 1e6:.1f} us per launch, {n * T * 299520 / dt / 1e12:.1f} TF algorithmic", flush=True)
so = ctypes.CDLL(L.LIB_PATH)
if not hasattr(so, "sepr_debug_gf5_trace"):
    sys.exit(0)      # not the trace build: timing only
buf = np.zeros((8, 512), dtype=np.uint64)
rc = so.sepr_debug_gf5_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
if "acc" in os.environ.get("SEPR_LIB_VARIANT", ""):
    # accumulator build: [wave][2 i] cycles, [2 i + 1] count of interval kind i (time since the previous stamp)
    names = {1: "barrier wait after CD / tile change", 2: "DMA issue (U)", 3: "up-projection work", 4: "barrier wait after U",
             5: "DMA issue (CD)", 6: "conv + GLU + split", 7: "down-projection", 8: "barrier wait after the last CD",
             9: "tile change work"}
    for wv in (0, 4):
        tot = sum(int(buf[wv, 2 * i]) for i in range(1, 10))
        print(f"--- wave {wv} (group {'A' if wv < 4 else 'B'}): {tot} cycles in the tile loop")
        for i in range(1, 10):
            cyc, cnt = int(buf[wv, 2 * i]), int(buf[wv, 2 * i + 1])
            if cnt:
                print(f"   {names[i]:38s} {cyc:9d} cycles  {100.0 * cyc / tot:5.1f} %   {cyc / cnt:8.0f} per occurrence (n={cnt})")
    sys.exit(0)
ids = (buf >> np.uint64(56)).astype(np.int64)
ts = (buf & np.uint64((1 << 56) - 1)).astype(np.int64)
base = min(int(ts[wv, 0]) for wv in range(8) if int(buf[wv, 511]) > 0)
for wv in (0, 4):
    cnt = int(buf[wv, 511])
    print(f"--- wave {wv} (group {'A' if wv < 4 else 'B'}), {cnt} stamps; s_memtime ticks relative to the first stamp")
    prev = None
    line = []
    for i in range(min(cnt, 140)):
        t = int(ts[wv, i]) - base
        d = 0 if prev is None else t - prev
        line.append(f"{ids[wv, i]}:{t}(+{d})")
        prev = t
        if ids[wv, i] in (7, 9):
            print("  " + " ".join(line)); line = []
    if line:
        print("  " + " ".join(line))
# summary: mean duration of each interval over the steady-state chunks
for wv in (0, 4):
    cnt = int(buf[wv, 511])
    seq = [(int(ids[wv, i]), int(ts[wv, i])) for i in range(cnt)]
    acc = {}
    for (ia, ta), (ib, tb) in zip(seq[:-1], seq[1:]):
        acc.setdefault((ia, ib), []).append(tb - ta)
    print(f"wave {wv}: " + "  ".join(f"{a}->{b}: {np.mean(v):.0f} (n={len(v)})" for (a, b), v in sorted(acc.items())))
