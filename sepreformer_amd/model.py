"""Drop-in ``Model``: the reference's ``nn.Module`` surface over the HIP separator path.

Boundary being mirrored:

* constructor kwargs == the ``config.model`` block of ``configs.yaml`` (reference ``model.py:14-21``,
  ``main.py:30``);
* ``forward(x: float32 [B,T]) -> (list[num_spks] of [B,T], list[num_stages] of list[num_spks] of [B,T])``
  (reference ``model.py:38-54``), auxiliary heads always evaluated, like the reference;
* ``state_dict()`` keys / shapes / order identical to the reference module tree, so
  ``load_state_dict`` of a reference checkpoint's ``model_state_dict`` works (``utils/util_engine.py:43``);
* attributes ``num_stages`` / ``num_spks`` (read by reference ``engine.py:52,88``); ``.to()``, ``.eval()``,
  ``.parameters()`` behave as for any ``nn.Module``.

The arithmetic is NOT here: ``forward`` hands the input to ``SeparatorEngine`` (HIP kernels through the
C ABI).  There is no CPU implementation in this package; a CPU tensor raises.  ``train()`` mode (dropout,
batch-statistics BatchNorm, autograd through the kernels) is a later row of SURVEY.md section 8f and raises
``NotImplementedError`` rather than silently computing eval-mode numbers.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch

from .config import SepConfig
from .engine import SeparatorEngine
from .pack import PRECISIONS, PackedModel
from .params import build_param_tree


DEFAULT_PRECISION = "bf16x3"


class Model(torch.nn.Module):
    def __init__(self, num_stages: int, num_spks: int, module_audio_enc: dict, module_feature_projector: dict,
                 module_separator: dict, module_output_layer: dict, module_audio_dec: dict,
                 per_level_split: bool = False, init_seed: Optional[int] = None, precision: Optional[str] = None):
        super().__init__()
        self.cfg = SepConfig.from_model_kwargs(num_stages, num_spks, module_audio_enc, module_feature_projector,
                                               module_separator, module_output_layer, module_audio_dec,
                                               per_level_split=per_level_split)
        self.num_stages = num_stages
        self.num_spks = num_spks
        self._kinds = build_param_tree(self, self.cfg, seed=init_seed)
        self._engine: Optional[SeparatorEngine] = None
        self._engine_key = None
        self.compute_aux = True    # the reference evaluates the aux heads in every forward (model.py:47-52)
        # latency mode: replay the forward from a captured hipGraph per input shape (engine.forward_graphed); the
        # returned tensors are then static buffers that the next same-shape call overwrites
        self.use_graphs = os.environ.get("SEPR_GRAPHS", "0") == "1"
        # throughput mode for batches: the batch as N independent pipelines on N streams (engine.forward_split)
        self.pipelines = int(os.environ.get("SEPR_PIPELINES", "1"))
        # projection arithmetic: "fp32" = exact f32 MFMA; "bf16x3" = split-fp32 on the bf16 MFMA (3 MFMAs per
        # product, ~100 dB agreement with fp32, 5x less matrix time).  Default from SEPR_PRECISION.
        self.precision = precision or os.environ.get("SEPR_PRECISION", DEFAULT_PRECISION)
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")

    # ---- weights -----------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, cfg: SepConfig, **kw) -> "Model":
        return cls(**cfg.model_kwargs(), per_level_split=cfg.per_level_split, **kw)

    def load_synthetic_(self, seed: int = 0) -> "Model":
        """Overwrite all tensors with the name-keyed O(1) synthetic weights (``synth.synth_state_dict``)."""
        from .synth import synth_state_dict
        self.load_state_dict(synth_state_dict(self.cfg, seed), strict=True)
        return self

    def _weights_key(self):
        # packed copies are caches keyed by (storage, version): optimizer steps / load_state_dict bump
        # ``_version``, ``.to()`` changes ``data_ptr``
        return tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())

    def engine(self) -> SeparatorEngine:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(
                "sepreformer_amd.Model computes on an MI355X (HIP) device only; move the module with "
                ".to('cuda'). There is deliberately no CPU fallback (the CPU restatement lives in oracle/ "
                "and is test infrastructure).")
        key = (dev, self.precision, self._weights_key())
        if self._engine is None or self._engine_key != key:
            sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            with torch.cuda.device(dev):
                self._engine = SeparatorEngine(self.cfg, PackedModel(self.cfg, sd, self.precision), dev)
            self._engine_key = key
        return self._engine

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if self.training:
            raise NotImplementedError(
                "train-mode forward/backward through the HIP kernels is not built yet (SURVEY.md section 8f-2); "
                "call .eval()")
        if x.dim() == 1:
            # the reference advertises [T] input but fails in GroupNorm for it
            raise RuntimeError("Expected [batch, samples] input")
        if not x.is_cuda:
            raise RuntimeError("input tensor is not on the HIP device (no CPU fallback exists)")
        eng = self.engine()
        with torch.cuda.device(x.device):
            if self.use_graphs:
                wav, aux = eng.forward_graphed(x.to(torch.float32), with_aux=self.compute_aux)
            elif self.pipelines > 1 and x.shape[0] >= 8 * self.pipelines:
                wav, aux = eng.forward_split(x.to(torch.float32), with_aux=self.compute_aux, parts=self.pipelines)
            else:
                wav, aux = eng.forward(x.to(torch.float32), with_aux=self.compute_aux)
        T = x.shape[-1]
        audio = [wav[s] for s in range(self.num_spks)]
        audio_aux = [[a[s][..., :T] for s in range(self.num_spks)] for a in aux]
        return audio, audio_aux

    @torch.no_grad()
    def separate(self, x: torch.Tensor) -> torch.Tensor:
        """Inference convenience: main outputs only, ``[S,B,T']`` in one tensor."""
        wav, _ = self.engine().forward(x, with_aux=False)
        return wav
