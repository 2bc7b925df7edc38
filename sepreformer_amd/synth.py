"""Deterministic synthetic weights and mixtures.

The pretrained checkpoint of the reference is a git-LFS pointer, so parity and
throughput are measured on synthesised weights.  Tensors are generated **by name** from a counter-based
generator (Philox keyed by ``(seed, crc32(name))``), so the build container (where the reference is
imported to produce ``tests/golden``) and the GPU box regenerate bit-identical tensors without shipping
them.  LayerScale / norm affine / BatchNorm statistics are drawn O(1): with the reference's 1e-5
LayerScale init every residual branch is invisible and bugs hide.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np
import torch

from .config import SepConfig
from .params import param_rows


def _gen(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def synth_tensor(name: str, shape, kind: str, fan_in: int, seed: int) -> torch.Tensor:
    g = _gen(seed, name)
    shp = tuple(shape)
    if kind == "proj":
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        a = g.uniform(-bound, bound, size=shp)
    elif kind in ("norm_w", "layer_scale", "bn_var"):
        a = g.uniform(0.5, 1.5, size=shp)
    elif kind in ("norm_b", "bn_mean"):
        a = g.normal(0.0, 0.1, size=shp)
    elif kind == "embed":
        a = g.normal(0.0, 1.0, size=shp)
    elif kind == "bn_count":
        return torch.tensor(1000, dtype=torch.long)
    else:
        raise KeyError(kind)
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shp).copy())


def synth_state_dict(cfg: SepConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Full ``state_dict`` (reference key names) of O(1)-scaled synthetic weights."""
    return {name: synth_tensor(name, shape, kind, fan, seed) for name, shape, kind, fan in param_rows(cfg)}


def synth_sources(batch: int, samples: int, seed: int = 1234, sr: int = 8000) -> np.ndarray:
    """``[batch, 2, samples]`` float32 pseudo-speech: white noise through a per-utterance 2-pole
    resonance with a slow (~4 Hz) amplitude envelope; utterance ``b`` uses seed ``seed + b``."""
    from scipy.signal import lfilter

    out = np.empty((batch, 2, samples), dtype=np.float32)
    t = np.arange(samples, dtype=np.float64) / sr
    for b in range(batch):
        g = _gen(seed + b, "mixture")
        for s in range(2):
            f0 = g.uniform(200.0, 1800.0)
            r = g.uniform(0.90, 0.98)
            w = 2.0 * math.pi * f0 / sr
            noise = g.normal(0.0, 1.0, size=samples)
            y = lfilter([1.0 - r], [1.0, -2.0 * r * math.cos(w), r * r], noise)
            env = 0.55 + 0.45 * np.sin(2.0 * math.pi * g.uniform(2.0, 6.0) * t + g.uniform(0, 2 * math.pi))
            y = y * env
            y = 0.05 * y / (np.sqrt(np.mean(y * y)) + 1e-12)
            out[b, s] = y.astype(np.float32)
    return out


def synth_mixture(batch: int, samples: int, seed: int = 1234) -> torch.Tensor:
    """``[batch, samples]`` float32 two-speaker mixtures (sum of ``synth_sources``)."""
    src = synth_sources(batch, samples, seed)
    return torch.from_numpy(src.sum(axis=1).astype(np.float32))
