#!/usr/bin/env python3
"""gemm_tn16_kernel (two bf16 operands, LDS-DMA ring + transposing LDS reads) against the register-staged gemm_tn_kernel<2>:
results vs an fp64 reference of the same bf16 inputs, and us per call (contraction + reduction) on the shapes of a Base training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sepreformer_amd import lib as L

lib = L.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(A, lda, B, ldb, M, N, K, G, cs, ws, wsb, acc=0):
    rc = lib.sepr_linear_wgrad_bf16(A.data_ptr(), lda, B.data_ptr(), ldb, G.data_ptr(), cs.data_ptr(), M, N, K, acc, ws.data_ptr(), wsb, st)
    assert rc == 0, rc


def setk(v):
    os.environ["SEPR_TN16"] = str(v)
    lib.sepr_knobs_reload()


def db(x, ref):
    e = (x.double() - ref).pow(2).sum().item()
    return 10 * torch.log10(ref.pow(2).sum() / max(e, 1e-300)).item()


torch.manual_seed(0)
ok = True
for (M, N, K, lda, ldb) in [(256, 128, 128, 128, 128), (4096, 768, 128, 768, 128), (4100, 128, 384, 128, 384), (8191, 256, 128, 256, 128), (33, 128, 128, 128, 128),
                            (1000, 128, 128, 384, 256), (64000, 768, 128, 768, 128), (31, 128, 128, 128, 128), (2049, 128, 256, 128, 256)]:
    A = (torch.randn(M, lda, device=dev) * (1 + torch.arange(lda, device=dev) % 7)).to(torch.bfloat16)
    B = torch.randn(M, ldb, device=dev).to(torch.bfloat16)
    refG = A[:, :N].double().t() @ B[:, :K].double()
    refs = A[:, :N].double().sum(0)
    wsb = lib.sepr_linear_wgrad_workspace(M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = {}
    for v in (1, 0):
        setk(v)
        G = torch.full((N, K), 7.0, device=dev); cs = torch.full((N,), 7.0, device=dev)
        run(A, lda, B, ldb, M, N, K, G, cs, ws, wsb)
        torch.cuda.synchronize()
        out[v] = (G, cs)
    d1, d0 = db(out[1][0], refG), db(out[0][0], refG)
    c1, c0 = db(out[1][1], refs), db(out[0][1], refs)
    good = d1 > 100 and c1 > 100
    ok = ok and good
    print(f"M={M:6d} N={N:4d} K={K:4d} lda={lda} ldb={ldb}: tn16 G {d1:6.1f} dB colsum {c1:6.1f} dB | staged G {d0:6.1f} dB colsum {c0:6.1f} dB  {'ok' if good else 'FAIL'}", flush=True)
    # accumulate form
    setk(1)
    G2 = out[1][0].clone(); cs2 = out[1][1].clone()
    run(A, lda, B, ldb, M, N, K, G2, cs2, ws, wsb, acc=1)
    torch.cuda.synchronize()
    if not (db(G2, 2 * refG) > 100 and db(cs2, 2 * refs) > 100):
        ok = False
        print("   accumulate FAIL")
print("PARITY", "OK" if ok else "FAILED")

for (M, N, K) in [(256000, 768, 128), (256000, 128, 384), (128000, 768, 128), (128000, 128, 384), (64000, 768, 128), (64000, 128, 384), (32000, 768, 128), (32000, 128, 384),
                  (16000, 768, 128), (16000, 128, 384), (256000, 256, 128), (8000, 768, 128)]:
    A = torch.randn(M, N, device=dev).to(torch.bfloat16)
    B = torch.randn(M, K, device=dev).to(torch.bfloat16)
    G = torch.empty(N, K, device=dev); cs = torch.empty(N, device=dev)
    wsb = lib.sepr_linear_wgrad_workspace(M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    res = {}
    for v in (0, 1, 0, 1):
        setk(v)
        for _ in range(3):
            run(A, N, B, K, M, N, K, G, cs, ws, wsb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(A, N, B, K, M, N, K, G, cs, ws, wsb)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    by = 2.0 * M * (N + K)
    print(f"M={M:6d} N={N:4d} K={K:4d}: staged {min(res[0]):7.1f} us ({by / min(res[0]) / 1e6:5.2f} TB/s)   tn16 {min(res[1]):7.1f} us ({by / min(res[1]) / 1e6:5.2f} TB/s)", flush=True)

# ---- fp32 operands (sepr_linear_wgrad, x3 = 2): SEPR_TN16=2 routes them through gemm_tnd_kernel<true, true> ----
print("fp32 operands, plain-bf16 arithmetic")
okf = True
for (M, N, K) in [(256, 128, 128), (4099, 384, 128), (8191, 128, 384), (33, 128, 128), (70000, 128, 128)]:
    A = torch.randn(M, N, device=dev) * (1 + torch.arange(N, device=dev) % 5)
    B = torch.randn(M, K, device=dev) + 0.25
    refG = A.bfloat16().double().t() @ B.bfloat16().double()
    refs = A.double().sum(0)
    wsb = lib.sepr_linear_wgrad_workspace(M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    o = {}
    for v in (2, 0):
        setk(v)
        G = torch.full((N, K), 7.0, device=dev); cs = torch.full((N,), 7.0, device=dev)
        rc = lib.sepr_linear_wgrad(A.data_ptr(), B.data_ptr(), G.data_ptr(), cs.data_ptr(), M, N, K, 0, 2, ws.data_ptr(), wsb, st)
        assert rc == 0
        torch.cuda.synchronize()
        o[v] = (G, cs)
    d2, d0 = db(o[2][0], refG), db(o[0][0], refG)
    c2, c0 = db(o[2][1], refs), db(o[0][1], refs)
    good = d2 > 100 and c2 > 40
    okf = okf and good
    print(f"M={M:6d} N={N:4d} K={K:4d}: dma G {d2:6.1f} dB colsum {c2:6.1f} dB | staged G {d0:6.1f} dB colsum {c0:6.1f} dB  {'ok' if good else 'FAIL'}", flush=True)
print("PARITY32", "OK" if okf else "FAILED")
for (M, N, K) in [(256000, 128, 128), (128000, 128, 128), (128000, 384, 128), (128000, 128, 384), (64000, 128, 128), (32000, 128, 128), (32000, 384, 128), (16000, 128, 128), (256000, 256, 128)]:
    A = torch.randn(M, N, device=dev)
    B = torch.randn(M, K, device=dev)
    G = torch.empty(N, K, device=dev); cs = torch.empty(N, device=dev)
    wsb = lib.sepr_linear_wgrad_workspace(M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    res = {}
    for v in (0, 2, 0, 2):
        setk(v)
        f = lambda: lib.sepr_linear_wgrad(A.data_ptr(), B.data_ptr(), G.data_ptr(), cs.data_ptr(), M, N, K, 0, 2, ws.data_ptr(), wsb, st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    by = 4.0 * M * (N + K)
    print(f"M={M:6d} N={N:4d} K={K:4d}: staged {min(res[0]):7.1f} us ({by / min(res[0]) / 1e6:5.2f} TB/s)   dma {min(res[2]):7.1f} us ({by / min(res[2]) / 1e6:5.2f} TB/s)", flush=True)
