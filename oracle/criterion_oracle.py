"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (PyTorch) restatement of the reference's permutation-invariant SI-SNR criteria
(``/root/reference/utils/implements/criterions.py``), same op sequence as the reference, batched:

* ``pit_sisnr_time``  - ``PIT_SISNR_time.__call__`` (:191-217): the training / validation loss;
* ``pit_sisnri``      - ``PIT_SISNRi.__call__`` (:232-260): the test-time SI-SNR improvement.

Only ``tests/`` may import this file.  Pinned by ``tests/golden/make_golden.py``, which runs the imported
reference classes (``torchaudio`` / ``mir_eval`` / ``loguru`` stubbed, they are not used by these two
classes) on seeded waveforms and stores inputs + outputs in ``tests/golden/criterion.npz``.

``dtype`` selects the arithmetic: ``torch.float32`` is the reference's own evaluation; ``torch.float64`` is the
exact-arithmetic value the device kernel (fp64 moments) is compared to at tight tolerance.
"""
from __future__ import annotations

from itertools import permutations
from typing import List, Tuple

import torch

Tensor = torch.Tensor


def _l2norm(mat: Tensor, keepdim: bool = False) -> Tensor:          # criterions.py:15-16
    return torch.norm(mat, dim=-1, keepdim=keepdim)


def _sisnr(a: Tensor, src: Tensor, eps: float) -> Tensor:
    """20 log10(eps + |alpha s~| / (|a~ - alpha s~| + eps)) per row (criterions.py:204-211 / :244-250)."""
    a_zm = a - torch.mean(a, dim=-1, keepdim=True)
    s_zm = src - torch.mean(src, dim=-1, keepdim=True)
    s_sc = torch.sum(a_zm * s_zm, dim=-1, keepdim=True) / (_l2norm(s_zm, keepdim=True) ** 2 + eps) * s_zm
    return 20 * torch.log10(eps + _l2norm(s_sc) / (_l2norm(a_zm - s_sc) + eps))


def pit_sisnr_time(estims: List[Tensor], targets: List[Tensor], eps: float = 1.0e-8,
                   dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (mean loss (scalar), per-utterance loss [B], best permutation [B,S] as target index per estimate)."""
    S = len(estims)
    est = [e.to(dtype) for e in estims]
    tgt = [t.to(dtype) for t in targets]
    perms = list(permutations(range(S)))
    pscore = []
    for p in perms:                                                  # :199-214
        tot = 0
        for s, t in enumerate(p):
            utt = -_sisnr(est[s], tgt[t], eps)
            tot = tot + torch.clamp(utt, min=-30)
        pscore.append(tot)
    pscore = torch.stack(pscore)                                     # [P, B]
    min_perutt, idx = torch.min(pscore, dim=0)                       # :215
    best = torch.tensor(perms)[idx]                                  # [B, S]
    return torch.sum(min_perutt) / est[0].shape[0], min_perutt, best


def pit_sisnri(estims: List[Tensor], targets: List[Tensor], mixture: Tensor, eps: float = 1.0e-15,
               dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (mean over utterances of the summed improvement, per-estimate improvements [B,S] of the best
    permutation, that permutation [B,S]).  The reference evaluates one utterance per call (:252 builds a
    tensor out of per-speaker scalars); this is the same computation row by row."""
    S = len(estims)
    est = [e.to(dtype) for e in estims]
    tgt = [t.to(dtype) for t in targets]
    x = mixture.to(dtype)
    perms = list(permutations(range(S)))
    pscore = []
    for p in perms:                                                  # :240-252
        per = []
        for s, t in enumerate(p):
            per.append(_sisnr(est[s], tgt[t], eps) - _sisnr(x, tgt[t], eps))
        pscore.append(torch.stack(per, dim=-1))                      # [B, S]
    pscore = torch.stack(pscore, dim=0)                              # [P, B, S]
    best_sum, idx = torch.max(pscore.sum(-1), dim=0)                 # :255
    B = x.shape[0]
    per_best = pscore[idx, torch.arange(B)]                          # [B, S]
    return torch.sum(best_sum) / B, per_best, torch.tensor(perms)[idx]
