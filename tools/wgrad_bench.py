#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient contraction (gemm_tn_kernel + tn_reduce_kernel) on the shapes of a Base training
step: us per call and TFLOP/s (algorithmic), against the HBM floor of reading both operands once."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd import lib as L

lib = L.load()
dev = torch.device("cuda:0")
shapes = [(256000, 768, 128), (256000, 128, 384), (128000, 768, 128), (128000, 128, 384), (32000, 768, 128), (32000, 128, 384), (32000, 384, 128), (32000, 128, 128), (64000, 768, 128), (64000, 128, 384),
          (16000, 768, 128), (8000, 768, 128), (4000, 384, 128), (2000, 128, 128)]
for M, N, K in shapes:
    A = torch.randn(M, N, device=dev)
    B = torch.randn(M, K, device=dev)
    G = torch.empty(N, K, device=dev)
    cs = torch.empty(N, device=dev)
    wsb = lib.sepr_linear_wgrad_workspace(M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    stats = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], 1).contiguous()
    NORM = os.environ.get("WGRAD_NORM", "0") == "1"      # the LayerNorm-prologue instantiation (sepr_linear_wgrad_norm)
    X3 = int(os.environ.get("WGRAD_X3", "1"))

    def run():
        if NORM:
            rc = lib.sepr_linear_wgrad_norm(A.data_ptr(), B.data_ptr(), stats.data_ptr(), G.data_ptr(), cs.data_ptr(), M, N, K, 0, X3, ws.data_ptr(), wsb, st)
        else:
            rc = lib.sepr_linear_wgrad(A.data_ptr(), B.data_ptr(), G.data_ptr(), cs.data_ptr(), M, N, K, 0, X3, ws.data_ptr(), wsb, st)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * N * K
    by = 4.0 * M * (N + K)
    print(f"M={M:6d} N={N:4d} K={K:4d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF  (HBM floor {by / 5e6:6.1f} us at 5 TB/s, ws {wsb / 1e6:.1f} MB)", flush=True)
