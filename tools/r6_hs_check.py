#!/usr/bin/env python3
"""The hidden-split GCFN / GLU-MLP form (gcfn_hs_kernel) against the oracle and - across processes - bitwise against the row-stationary kernels
it replaces for small launches (SEPR_GF_HS=0): prints, per case, the agreement with the oracle and a sha256 of the output; on a mismatch, where
the rows differ.  tests/test_gpu_parity.py::test_gcfn_hidden_split_bitwise runs it under both settings and compares the lines.
R6_HS_TIME=1 adds per-launch times at sizes around the form boundaries (profiles/r06_gcfn_hidden_split.txt)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_parity import gpu_model, rnd, orc   # noqa: E402
from sepreformer_amd.synth import synth_mixture          # noqa: E402

m, sd = gpu_model("SepReformer_Base_WSJ0", "bf16x3")
eng = m.engine()
eng.prepare(8, 2400, 2400)
# (n sequences, T frames): tile ends on / next to sequence ends, every tile size's last full launch and the first launch past it
CASES = ((1, 1), (2, 37), (1, 30), (1, 31), (3, 300), (5, 2), (2, 127), (1, 253), (1, 2000), (2, 2300), (1, 7680), (1, 7681), (1, 9000),
         (3, 3925), (1, 11777), (1, 13000), (2, 7936), (1, 15873))
for n, T in CASES:
    x = rnd(n, T, m.cfg.feat, seed=T)
    y = eng.gcfn(x.cuda(), eng.pk.enc_stages[0]["g"][0][1], n, T).cpu()
    ref = orc.gcfn(sd, "separator.enc_stages.0.g_block_1.block.gcfn", x)
    db = orc.agreement_db(y, ref)
    line = f"gcfn {n} {T} {db:.1f} {hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]}"
    if db < 80:
        d = (y - ref).abs().reshape(n * T, -1)
        bad = (d.max(1).values > 1e-3).nonzero().flatten().tolist()
        cols = (d.max(0).values > 1e-3).nonzero().flatten().tolist()
        line += f" bad_rows[{len(bad)}]={bad[:24]} bad_cols[{len(cols)}]={cols[:40]}"
    print(line, flush=True)
# the CLA block (cla_tail_hs_kernel: 32- / 64-frame tiles; boundaries 8192 / 16384 rows)
for n, T in ((1, 1), (2, 37), (1, 32), (1, 33), (3, 300), (1, 2000), (1, 8192), (1, 8193), (2, 8000), (1, 16384), (1, 16385)):
    x = rnd(n, T, m.cfg.feat, seed=T + 1)
    y = eng.cla(x.cuda(), eng.pk.enc_stages[0]["l"][0][0], n, T).cpu()
    db = orc.agreement_db(y, orc.cla(sd, "separator.enc_stages.0.l_block_1.block.cla", x))
    print(f"cla {n} {T} {db:.1f} {hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]}", flush=True)


# the EGA block (output-split q / k / v and gate launches; pooling factors 1 .. 16; boundaries 8192 / 16384 rows)
for n, T, Tp in ((1, 25, 25), (2, 50, 25), (1, 2000, 250), (1, 8000, 1000), (1, 8192, 512), (1, 8208, 513), (2, 8000, 1000), (1, 16384, 1024), (1, 16400, 1025)):
    x = rnd(n, T, m.cfg.feat, seed=T + 3)
    y = eng.ega(x.cuda(), eng.pk.enc_stages[0]["g"][0][0], n, T, Tp).cpu()
    print(f"ega {n} {T} {Tp} {hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]}", flush=True)

# the speaker attention (spk_hs_kernel: 16- / 32-frame tiles; boundaries 4096 / 8192 frames); n = mixtures x 2 speakers
w_att = eng.pk.dec_stages[0]["spk"][0][0]
for n, T in ((2, 1), (4, 33), (2, 4096), (2, 4097), (4, 2500), (2, 8000), (2, 8193)):
    x = rnd(n, T, m.cfg.feat, seed=T + 2)
    y = eng.spkattn(x.cuda(), w_att, n, T).cpu()
    print(f"spk {n} {T} {hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]}", flush=True)


def flat(o):
    if torch.is_tensor(o):
        yield o
    else:
        for e in o:
            yield from flat(e)


# the whole separator on one utterance: SpkSplit's and OutputLayer's GLU-MLP launches take the PLAIN instantiations
for L in (4000, 32000, 47001):
    out = m(synth_mixture(1, L, seed=L).cuda())
    hh = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in flat(out))).hexdigest()[:16]
    print(f"model 1 {L} {hh}", flush=True)

if os.environ.get("R6_HS_TIME"):
    # per-launch time of the GCFN block at launch sizes around the form boundaries (200 back-to-back launches between two HIP events)
    for M in (2000, 7680, 8000, 11776, 12000, 15000, 15872, 16000):
        x = rnd(1, M, m.cfg.feat, seed=M).cuda()
        wgt = eng.pk.enc_stages[0]["g"][0][1]
        for _ in range(20):
            eng.gcfn(x, wgt, 1, M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            eng.gcfn(x, wgt, 1, M)
        e1.record()
        torch.cuda.synchronize()
        print(f"time M={M}: {e0.elapsed_time(e1) * 5:.1f} us per launch (includes launch gaps)", flush=True)
