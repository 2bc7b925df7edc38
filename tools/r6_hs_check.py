#!/usr/bin/env python3
"""Round 6: the hidden-split GCFN form (gcfn_hs_kernel, SEPR_GF_HS=2|4) against the oracle AND bitwise against the batched kernel
(SEPR_GF_HS=0, own process): prints per-shape agreement, a sha256 of every output and - on a mismatch - where the rows differ."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_parity import gpu_model, rnd, orc   # noqa: E402

m, sd = gpu_model("SepReformer_Base_WSJ0", "bf16x3")
eng = m.engine()
eng.prepare(8, 2400, 2400)
dump = os.environ.get("R6_HS_DUMP")
for n, T in ((1, 1), (2, 37), (1, 30), (1, 31), (3, 300), (5, 2), (2, 127), (1, 253), (1, 2000), (2, 2300), (1, 7680), (1, 7681)):
    x = rnd(n, T, m.cfg.feat, seed=T)
    y = eng.gcfn(x.cuda(), eng.pk.enc_stages[0]["g"][0][1], n, T).cpu()
    ref = orc.gcfn(sd, "separator.enc_stages.0.g_block_1.block.gcfn", x)
    db = orc.agreement_db(y, ref)
    line = f"{n} {T} {db:.1f} {hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]}"
    if db < 80:
        d = (y - ref).abs().reshape(n * T, -1)
        bad = (d.max(1).values > 1e-3).nonzero().flatten().tolist()
        cols = (d.max(0).values > 1e-3).nonzero().flatten().tolist()
        line += f" bad_rows[{len(bad)}]={bad[:24]} bad_cols[{len(cols)}]={cols[:40]}"
    print(line, flush=True)
    if dump:
        np.save(os.path.join(dump, f"hs_{n}_{T}.npy"), y.numpy())
