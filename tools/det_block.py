#!/usr/bin/env python3
"""Run-to-run determinism of single blocks' backward at full size; prints which gradient tensors differ."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd import lib as L
from sepreformer_amd.config import VARIANTS
from sepreformer_amd.synth import synth_state_dict
from sepreformer_amd.train_engine import TrainEngine
from sepreformer_amd.train_pack import GradBuffer, TrainPack

dev = torch.device("cuda:0")
cfg = dataclasses.replace(VARIANTS["SepReformer_Base_WSJ0"], dropout=0.0)
sd = {k: v.to(dev) for k, v in synth_state_dict(cfg, 0).items()}
gb = GradBuffer(cfg, dev)
tp = TrainPack(cfg, sd, gb, "bf16x3")
eng = TrainEngine(cfg, dev)
F = cfg.feat
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in ((64000, 768, 128), (64000, 128, 384), (64000, 256, 128)):
    a, b = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    ws = torch.empty(int(lib.sepr_linear_wgrad_workspace(M, N, K)) + 256, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(4):
        G = torch.empty(N, K, device=dev)
        L.check(lib.sepr_linear_wgrad(a.data_ptr(), b.data_ptr(), G.data_ptr(), None, M, N, K, 0, 1, ws.data_ptr(), ws.numel(), st), "wgrad")
        outs.append(G)
    print(f"wgrad {M}x{N}x{K}: repeats equal:", [bool(torch.equal(outs[0], o)) for o in outs[1:]])
for kind, w, n, T in (("gcfn", tp.gcfn[0], 8, 8000), ("cla", tp.cla[0], 8, 8000), ("spk", tp.spk[0], 8, 8000), ("gcfn", tp.gcfn[0], 2, 1000)):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, T, F, generator=g).to(dev)
    dy = torch.randn(n, T, F, generator=g).to(dev)
    res = []
    for rep in range(3):
        gb.flat.zero_()
        y, rec = eng.block_fwd(kind, x, w, n, T)
        dx = eng.block_bwd(rec, dy)
        torch.cuda.synchronize()
        res.append((y.clone(), dx.clone(), gb.flat.clone()))
    for rep in (1, 2):
        names = [k for k in gb.offsets if not torch.equal(res[0][2][gb.offsets[k][0]:gb.offsets[k][0] + 1 + max(0, torch.tensor(gb.offsets[k][1]).prod().item() - 1)],
                                                           res[rep][2][gb.offsets[k][0]:gb.offsets[k][0] + 1 + max(0, torch.tensor(gb.offsets[k][1]).prod().item() - 1)])]
        print(f"{kind} n={n} T={T} rep{rep}: y equal {bool(torch.equal(res[0][0], res[rep][0]))}, dx equal {bool(torch.equal(res[0][1], res[rep][1]))}, differing grads: {[k.split('.', 4)[-1] for k in names]}")
