"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Train-mode restatement of the reference's training step (``/root/reference/models/SepReformer_Base_WSJ0/engine.py:50-83``)
on the CPU: the forward of ``oracle/sepreformer_oracle.py`` with BatchNorm in training mode (batch statistics over
(batch, frames), running statistics updated with momentum 0.1; reference ``modules/network.py:167,183``,
``modules/module.py:69,75``) and dropout p = 0, differentiated by ``torch.autograd`` - the same mechanism the reference
uses.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may import this file.

Pinning: ``tests/golden/make_train_golden.py`` runs the IMPORTED reference ``Model`` in ``train()`` mode (dropout_rate 0) with
the reference's own criteria, back-propagates, and checks every parameter gradient, the loss terms and the updated running
statistics of this oracle against it (``tests/golden/PINNING_train.json``); compact per-tensor summaries of the reference
gradients are committed in ``tests/golden/train_tiny.npz`` and re-checked by ``tests/test_train_oracle.py``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import criterion_oracle as co
from . import sepreformer_oracle as orc

Tensor = torch.Tensor


def leaf_state(sd: Dict[str, Tensor], dtype=torch.float32) -> Dict[str, Tensor]:
    """Copy of a state_dict whose floating parameters are autograd leaves (buffers stay plain tensors)."""
    out = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            out[k] = v.clone()
        elif k.endswith("running_mean") or k.endswith("running_var"):
            out[k] = v.clone().to(dtype)
        else:
            out[k] = v.clone().to(dtype).requires_grad_(True)
    return out


def model_forward_train(sd: Dict[str, Tensor], cfg, x: Tensor):
    """``Model.forward`` under ``model.train()`` with dropout p = 0 (model.py:38-54).  Updates ``sd``'s running statistics."""
    orc.BN_TRAINING = True
    try:
        return orc.model_forward.__wrapped__(sd, cfg, x)
    finally:
        orc.BN_TRAINING = False


def train_loss(audio: List[Tensor], audio_aux: List[List[Tensor]], src: List[Tensor], alpha: float = 0.4, frame_len: int = 512,
               frame_shift: int = 128) -> Tuple[Tensor, Tensor, List[Tensor]]:
    """Loss of reference ``engine.py:66-74``: ((1 - alpha) PIT_SISNR_time(main) + alpha mean_i PIT_SISNR_mag(aux_i)) / num_spks."""
    S = len(audio)
    l_time = co.pit_sisnr_time(audio, src)[0]
    l_mag = [co.pit_sisnr_mag(a, src, frame_len, frame_shift)[0] for a in audio_aux]
    loss = ((1.0 - alpha) * l_time + alpha * sum(l_mag) / len(l_mag)) / S
    return loss, l_time, l_mag
