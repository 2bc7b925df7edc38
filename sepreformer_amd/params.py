"""Parameter tree with the reference's ``state_dict`` contract.

The separator's arithmetic lives in HIP kernels, so the PyTorch side only has to *own* the tensors under
the names a reference checkpoint uses (``model_state_dict`` written by reference
``utils/util_engine.py:96-111`` and read back with ``load_state_dict(strict=False)`` at ``:43``).
Instead of mirroring the reference's class hierarchy, the tree is generated from a flat, declarative
list of ``(dotted.name, shape, kind)`` rows; ``ParamNode`` is a behaviour-less ``nn.Module`` container.

The row order reproduces the reference's registration order so ``list(state_dict())`` is identical
(checked against ``tests/golden/state_dict_*.json``).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Tuple

import torch

from .config import SepConfig

# kind -> (is_buffer, init family)
KINDS = {
    "proj": False,        # Linear / Conv weight or bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    "norm_w": False,      # LayerNorm / GroupNorm / BatchNorm weight: ones
    "norm_b": False,      # ... bias: zeros
    "embed": False,       # Embedding: N(0, 1)
    "layer_scale": False, # LayerScale: 1e-5 (reference modules/network.py:8-18)
    "bn_mean": True,      # BatchNorm running_mean: zeros
    "bn_var": True,       # BatchNorm running_var: ones
    "bn_count": True,     # BatchNorm num_batches_tracked: int64 scalar
}

Row = Tuple[str, Tuple[int, ...], str, int]  # name, shape, kind, fan_in


# Bumped whenever a tensor ATTRIBUTE of the tree is (re)bound - ``node.weight = nn.Parameter(...)``,
# ``load_state_dict(..., assign=True)``, parametrizations - i.e. whenever a cached list of the tree's tensor objects may
# hold stale objects.  (In-place updates and ``.data`` re-assignments keep the objects: Model._weights_key sees those.)
_MUTATIONS = [0]


def mutation_epoch() -> int:
    return _MUTATIONS[0]


class ParamNode(torch.nn.Module):
    """Pure container: holds parameters/buffers/children, has no forward of its own."""

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor) and not self.__dict__.get("_is_replica", False):
            _MUTATIONS[0] += 1       # (replicas of torch.nn.parallel.replicate rebind plain tensors on throw-away copies)
        super().__setattr__(name, value)

    def register_parameter(self, name, param):
        _MUTATIONS[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        _MUTATIONS[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)

    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("ParamNode only stores tensors; call Model.forward")


def _rows_linear(out: List[Row], p: str, n_out: int, n_in: int) -> None:
    out.append((p + ".weight", (n_out, n_in), "proj", n_in))
    out.append((p + ".bias", (n_out,), "proj", n_in))


def _rows_conv(out: List[Row], p: str, n_out: int, n_in_per_group: int, k: int, bias: bool = True) -> None:
    fan = n_in_per_group * k
    out.append((p + ".weight", (n_out, n_in_per_group, k), "proj", fan))
    if bias:
        out.append((p + ".bias", (n_out,), "proj", fan))


def _rows_norm(out: List[Row], p: str, c: int) -> None:
    out.append((p + ".weight", (c,), "norm_w", 0))
    out.append((p + ".bias", (c,), "norm_b", 0))


def _rows_bn(out: List[Row], p: str, c: int) -> None:
    _rows_norm(out, p, c)
    out.append((p + ".running_mean", (c,), "bn_mean", 0))
    out.append((p + ".running_var", (c,), "bn_var", 0))
    out.append((p + ".num_batches_tracked", (), "bn_count", 0))


def _rows_mha(out: List[Row], p: str, F: int) -> None:
    # reference modules/network.py:76-88
    _rows_norm(out, p + ".layer_norm", F)
    for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
        _rows_linear(out, f"{p}.{nm}", F, F)
    out.append((p + ".Layer_scale.layer_scale", (1, 1, F), "layer_scale", 0))


def _rows_gcfn(out: List[Row], p: str, F: int) -> None:
    # reference modules/network.py:47-58
    _rows_norm(out, p + ".net1.0", F)
    _rows_linear(out, p + ".net1.1", 6 * F, F)
    _rows_conv(out, p + ".depthwise", 6 * F, 1, 3)
    _rows_linear(out, p + ".net2.2", F, 3 * F)
    out.append((p + ".Layer_scale.layer_scale", (1, 1, F), "layer_scale", 0))


def _rows_global(out: List[Row], p: str, F: int) -> None:
    # reference modules/network.py:127-136,190-196
    _rows_mha(out, p + ".block.ega.block.self_attn", F)
    _rows_norm(out, p + ".block.ega.block.linear.0", F)
    _rows_linear(out, p + ".block.ega.block.linear.1", F, F)
    _rows_gcfn(out, p + ".block.gcfn", F)


def _rows_local(out: List[Row], p: str, F: int, k: int) -> None:
    # reference modules/network.py:160-172,213-218
    c = p + ".block.cla"
    _rows_norm(out, c + ".layer_norm", F)
    _rows_linear(out, c + ".linear1", 2 * F, F)
    _rows_conv(out, c + ".dw_conv_1d", F, 1, k)
    _rows_linear(out, c + ".linear2", 2 * F, F)
    _rows_bn(out, c + ".BN", 2 * F)
    _rows_linear(out, c + ".linear3.1", F, 2 * F)
    out.append((c + ".Layer_scale.layer_scale", (1, 1, F), "layer_scale", 0))
    _rows_gcfn(out, p + ".block.gcfn", F)


def _rows_enc_stage(out: List[Row], p: str, cfg: SepConfig, down: bool) -> None:
    # reference modules/module.py:59-86
    F = cfg.feat
    _rows_global(out, p + ".g_block_1", F)
    _rows_local(out, p + ".l_block_1", F, cfg.cla_kernel)
    _rows_global(out, p + ".g_block_2", F)
    _rows_local(out, p + ".l_block_2", F, cfg.cla_kernel)
    if down:
        _rows_conv(out, p + ".downconv.down_conv", F, 1, cfg.down_kernel)
        _rows_bn(out, p + ".downconv.BN", F)


def _rows_split(out: List[Row], p: str, cfg: SepConfig) -> None:
    # reference modules/module.py:110-118
    F, S = cfg.feat, cfg.num_spks
    _rows_conv(out, p + ".linear.0", 4 * F * S, F, 1)
    _rows_conv(out, p + ".linear.2", F * S, 2 * F * S, 1)
    _rows_norm(out, p + ".norm", F)


def _rows_dec_stage(out: List[Row], p: str, cfg: SepConfig) -> None:
    # reference modules/module.py:127-143
    F = cfg.feat
    for i in (1, 2, 3):
        _rows_global(out, f"{p}.g_block_{i}", F)
        _rows_local(out, f"{p}.l_block_{i}", F, cfg.cla_kernel)
        _rows_mha(out, f"{p}.spk_attn_{i}.self_attn", F)
        _rows_gcfn(out, f"{p}.spk_attn_{i}.feed_forward", F)


def _rows_out_layer(out: List[Row], p: str, cfg: SepConfig) -> None:
    # reference modules/module.py:238-247 (Masking holds no parameters with concat_opt=None)
    _rows_linear(out, p + ".end_conv1x1.0", 4 * cfg.feat, cfg.feat)
    _rows_linear(out, p + ".end_conv1x1.2", cfg.enc_channels, 2 * cfg.feat)


def param_rows(cfg: SepConfig) -> List[Row]:
    """Every tensor of the model, in the reference's ``state_dict`` order (reference model.py:22-36)."""
    rows: List[Row] = []
    N, F, R = cfg.enc_channels, cfg.feat, cfg.num_stages
    _rows_conv(rows, "audio_encoder.conv1d", N, 1, cfg.enc_kernel, bias=False)
    _rows_norm(rows, "feature_projector.norm", N)
    _rows_conv(rows, "feature_projector.conv1d", F, N, 1, bias=False)
    rows.append(("separator.pos_emb.pe_k.weight", (2 * cfg.maxlen, cfg.dk), "embed", 0))
    for i in range(R):
        _rows_enc_stage(rows, f"separator.enc_stages.{i}", cfg, down=True)
    _rows_enc_stage(rows, "separator.bottleneck_G", cfg, down=False)
    if cfg.per_level_split:
        for i in range(R + 1):
            _rows_split(rows, f"separator.spk_split_blocks.{i}", cfg)
    else:
        _rows_split(rows, "separator.spk_split_block", cfg)
    # the reference appends simple_fusion[i] and dec_stages[i] in one loop, but state_dict walks one
    # ModuleList after the other (modules/module.py:184-188)
    for i in range(R):
        _rows_conv(rows, f"separator.simple_fusion.{i}", F, 2 * F, 1)
    for i in range(R):
        _rows_dec_stage(rows, f"separator.dec_stages.{i}", cfg)
    _rows_out_layer(rows, "out_layer", cfg)
    # ConvTranspose1d weight is (in_channels, out_channels/groups, k); torch takes fan_in from dim 1
    rows.append(("audio_decoder.weight", (N, 1, cfg.enc_kernel), "proj", cfg.enc_kernel))
    for i in range(R):
        _rows_out_layer(rows, f"out_layer_bn.{i}", cfg)
    for i in range(R):
        rows.append((f"decoder_bn.{i}.weight", (N, 1, cfg.enc_kernel), "proj", cfg.enc_kernel))
    return rows


def _default_tensor(shape, kind: str, fan_in: int, gen: torch.Generator) -> torch.Tensor:
    if kind == "proj":
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound
    if kind in ("norm_w", "bn_var"):
        return torch.ones(shape)
    if kind in ("norm_b", "bn_mean"):
        return torch.zeros(shape)
    if kind == "embed":
        return torch.randn(shape, generator=gen)
    if kind == "layer_scale":
        return torch.full(shape, 1.0e-5)
    if kind == "bn_count":
        return torch.zeros((), dtype=torch.long)
    raise KeyError(kind)


def build_param_tree(root: torch.nn.Module, cfg: SepConfig, seed: int | None = None) -> Dict[str, str]:
    """Attach the whole parameter tree under ``root``; returns ``{name: kind}``."""
    gen = torch.Generator()
    gen.manual_seed(torch.seed() if seed is None else seed)
    kinds: Dict[str, str] = {}
    for name, shape, kind, fan_in in param_rows(cfg):
        *path, leaf = name.split(".")
        node = root
        for part in path:
            child = node._modules.get(part)
            if child is None:
                child = ParamNode()
                node.add_module(part, child)
            node = child
        t = _default_tensor(shape, kind, fan_in, gen)
        if KINDS[kind]:
            node.register_buffer(leaf, t)
        else:
            node.register_parameter(leaf, torch.nn.Parameter(t))
        kinds[name] = kind
    return kinds


def count_parameters(cfg: SepConfig, include_aux: bool = True) -> int:
    n = 0
    for name, shape, kind, _ in param_rows(cfg):
        if KINDS[kind]:
            continue
        if not include_aux and (name.startswith("out_layer_bn.") or name.startswith("decoder_bn.")):
            continue
        n += int(math.prod(shape)) if shape else 1
    return n


def rows_for(cfg: SepConfig, predicate: Callable[[str], bool]) -> List[Row]:
    return [r for r in param_rows(cfg) if predicate(r[0])]
