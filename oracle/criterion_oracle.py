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


def stft_kernel(frame_len: int, frame_hop: int) -> Tensor:
    """STFTBase._init_kernel, criterions.py:43-61 ('hann' window): ``[frame_len + 2, 1, frame_len]``."""
    N = frame_len
    W = torch.hann_window(frame_len)                                  # :48
    if N // 4 == frame_hop:                                           # :49-51
        W = (2 / 3) ** 0.5 * W
    elif N // 2 == frame_hop:                                         # :52-53
        W = W ** 0.5
    S = 0.5 * (N * N / frame_hop) ** 0.5                              # :54
    K = torch.fft.rfft(torch.eye(N) / S, dim=1)[:frame_len]           # :57
    K = torch.stack((torch.real(K), torch.imag(K)), dim=2)            # :58
    K = torch.transpose(K, 0, 2) * W                                  # :59
    return torch.reshape(K, (N + 2, 1, frame_len))                    # :60


def stft_mag(x: Tensor, K: Tensor, frame_shift: int) -> Tensor:
    """STFT.forward for ``[N, S]`` input, magnitude only (criterions.py:88-113)."""
    from math import ceil
    n_frame = ceil(x.shape[-1] / frame_shift)                         # :90
    len_padded = n_frame * frame_shift
    x = torch.cat((x, torch.zeros(x.shape[0], len_padded - x.shape[-1], dtype=x.dtype)), dim=-1)   # :94
    c = torch.nn.functional.conv1d(x.unsqueeze(1), K.to(x.dtype), stride=frame_shift, padding=0)    # :97
    r, i = torch.chunk(c, 2, dim=1)                                   # :99
    return (r ** 2 + i ** 2 + 1.0e-10) ** 0.5                         # :112


def pit_sisnr_mag(estims: List[Tensor], targets: List[Tensor], frame_length: int = 512, frame_shift: int = 128,
                  eps: float = 1.0e-12, dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Tensor]:
    """PIT_SISNR_mag.__call__ (criterions.py:148-176, mel_opt False) -> (mean loss, per-utterance loss [B], perm [B,S])."""
    S = len(estims)
    K = stft_kernel(frame_length, frame_shift)
    est = [e.to(dtype) for e in estims]
    tgt = [t.to(dtype) for t in targets]
    perms = list(permutations(range(S)))
    pscore = []
    for p in perms:
        tot = 0
        for s, t in enumerate(p):                                     # :154-166
            mix_zm = est[s] - torch.mean(est[s], dim=-1, keepdim=True)
            src_zm = tgt[t] - torch.mean(tgt[t], dim=-1, keepdim=True)
            scale = torch.sum(mix_zm * src_zm, dim=-1, keepdim=True) / (_l2norm(src_zm, keepdim=True) ** 2 + eps)
            src_zm = torch.clamp(scale, min=1e-2) * src_zm
            m_mix, m_src = stft_mag(mix_zm, K, frame_shift), stft_mag(src_zm, K, frame_shift)
            tot = tot + (-20 * torch.log10(eps + _l2norm(_l2norm(m_src)) / (_l2norm(_l2norm(m_mix - m_src)) + eps)))
        pscore.append(tot)
    pscore = torch.stack(pscore)
    min_perutt, idx = torch.min(pscore, dim=0)
    return torch.sum(min_perutt) / est[0].shape[0], min_perutt, torch.tensor(perms)[idx]
