#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd import lib as L
dev = torch.device("cuda:0")
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in ((64000, 768, 128), (64000, 256, 128), (20000, 768, 128), (512000, 128, 128)):
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(M, N, generator=g).to(dev), (torch.randn(M, K, generator=g) * 2 + 0.5).to(dev)
    mean = b.mean(1)
    rstd = 1.0 / torch.sqrt(b.var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1).contiguous()
    ws = torch.empty(int(lib.sepr_linear_wgrad_workspace(M, N, K)) + 256, dtype=torch.uint8, device=dev)
    want = (a.double().t() @ ((b.double() - mean.double()[:, None]) * rstd.double()[:, None])).float()
    for x3 in (0, 1):
        outs = []
        for _ in range(4):
            G = torch.empty(N, K, device=dev)
            L.check(lib.sepr_linear_wgrad_norm(a.data_ptr(), b.data_ptr(), stats.data_ptr(), G.data_ptr(), None, M, N, K, 0, x3, ws.data_ptr(), ws.numel(), st), "wgrad")
            outs.append(G)
        err = [float((o - want).abs().max() / want.abs().max()) for o in outs]
        print(f"wgrad_norm {M}x{N}x{K} x3={x3}: repeats equal {[bool(torch.equal(outs[0], o)) for o in outs[1:]]} rel err {['%.1e' % e for e in err]}")
        if not torch.equal(outs[0], outs[1]):
            d = (outs[0] - outs[1]).abs()
            nz = (d > 0).nonzero()
            print("   differing entries:", nz.shape[0], "rows(n) range", int(nz[:, 0].min()), int(nz[:, 0].max()), "cols(k) range", int(nz[:, 1].min()), int(nz[:, 1].max()), "max", float(d.max()))
