#!/usr/bin/env python3
"""Fault probe 1 (round 5): the weight-gradient contraction's GENERAL loader with two workgroups per CU.

Round 3 found wrong, run-to-run different values (even columns k of the upper half of a B tile) whenever the bf16 instantiations of the
per-row-conditional loader normalised their B rows AND two workgroups shared a CU; the product keeps that loader at one workgroup per
CU with a 16 KB LDS pad.  This script drives the SAME kernel through the public entry point with the pad overridden:

    SEPR_TN_FORCE_GEN=1 SEPR_TN_GEN_PAD=<bytes> [SEPR_LIB_VARIANT=<tag>] python tools/probe/tn_fault.py

and prints, per shape and arithmetic: repeat equality, error vs fp64, and WHERE the repeats differ (k parity, k range inside the tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sepreformer_amd import lib as L
dev = torch.device("cuda:0")
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
tag = f"[gen={os.environ.get('SEPR_TN_FORCE_GEN','0')} pad={os.environ.get('SEPR_TN_GEN_PAD','16384')} lib={os.environ.get('SEPR_LIB_VARIANT','default')}]"
for (M, N, K) in ((64000, 768, 128), (128000, 128, 128), (20000, 768, 128)):
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(M, N, generator=g).to(dev), (torch.randn(M, K, generator=g) * 2 + 0.5).to(dev)
    mean = b.mean(1)
    rstd = 1.0 / torch.sqrt(b.var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1).contiguous()
    ws = torch.empty(int(lib.sepr_linear_wgrad_workspace(M, N, K)) + 256, dtype=torch.uint8, device=dev)
    want = (a.double().t() @ ((b.double() - mean.double()[:, None]) * rstd.double()[:, None])).float()
    for x3 in (0, 1, 2):
        outs = []
        for _ in range(6):
            G = torch.empty(N, K, device=dev)
            L.check(lib.sepr_linear_wgrad_norm(a.data_ptr(), b.data_ptr(), stats.data_ptr(), G.data_ptr(), None, M, N, K, 0, x3, ws.data_ptr(), ws.numel(), st), "wgrad")
            outs.append(G)
        torch.cuda.synchronize()
        err = max(float((o - want).abs().max() / want.abs().max()) for o in outs)
        eq = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
        line = f"{tag} wgrad_norm {M}x{N}x{K} x3={x3}: repeats_equal={eq} max_rel_err={err:.2e}"
        if not eq:
            d = torch.zeros_like(outs[0], dtype=torch.bool)
            for o in outs[1:]:
                d |= (o != outs[0])
            nz = d.nonzero()
            kk, nn = nz[:, 1], nz[:, 0]
            line += (f" | differing {nz.shape[0]} of {d.numel()}: k even {int((kk % 2 == 0).sum())} odd {int((kk % 2 == 1).sum())}; "
                     f"k in [0,64) {int((kk % 128 < 64).sum())} [64,128) {int((kk % 128 >= 64).sum())}; n%4 hist {[int((nn % 4 == i).sum()) for i in range(4)]}; "
                     f"n range {int(nn.min())}-{int(nn.max())}")
        print(line, flush=True)
