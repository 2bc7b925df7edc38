"""Model hyper-parameters of the separator path.

The reference passes the ``config.model`` block of ``configs.yaml`` straight into
``Model(**cfg)`` (reference ``models/SepReformer_Base_WSJ0/main.py:30``,
``models/SepReformer_Base_WSJ0/model.py:14-21``).  ``SepConfig.from_model_kwargs`` takes the same
nested dict and flattens it into the handful of integers the HIP path needs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, asdict
from typing import Any, Dict

import yaml


@dataclass(frozen=True)
class SepConfig:
    num_stages: int = 4          # R   U-Net depth (configs.yaml:31)
    num_spks: int = 2            # S
    enc_channels: int = 256      # N   encoder filters (configs.yaml:35)
    enc_kernel: int = 16         # encoder / decoder taps (configs.yaml:36)
    enc_stride: int = 4          # encoder / decoder hop (configs.yaml:37)
    feat: int = 128              # F   separator width (configs.yaml:44)
    heads: int = 8               # H
    maxlen: int = 2000           # relative-position clamp (configs.yaml:51)
    cla_kernel: int = 65         # CLA depthwise taps (configs.yaml:62)
    down_kernel: int = 5         # DownConv taps (configs.yaml:66)
    dropout: float = 0.05        # train-only
    per_level_split: bool = False  # Large_DM_WHAM keeps one SpkSplitStage per level

    @property
    def dk(self) -> int:
        return self.feat // self.heads

    @property
    def frame_multiple(self) -> int:
        return 2 ** self.num_stages

    def frames(self, samples: int) -> int:
        """Encoder frames for ``samples`` input samples (Conv1d k/stride, no padding)."""
        return (samples - self.enc_kernel) // self.enc_stride + 1

    def padded_frames(self, frames: int) -> int:
        """Separator.pad_signal (reference modules/module.py:220-234): pad to a multiple of 2**R,
        nothing added when already a multiple."""
        m = self.frame_multiple
        return frames if frames % m == 0 else (frames // m + 1) * m

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @staticmethod
    def from_model_kwargs(num_stages: int, num_spks: int, module_audio_enc: dict,
                          module_feature_projector: dict, module_separator: dict,
                          module_output_layer: dict, module_audio_dec: dict,
                          per_level_split: bool = False) -> "SepConfig":
        sep = module_separator
        enc_stage = sep["enc_stage"]
        dec_stage = sep["dec_stage"]
        feat = int(module_feature_projector["out_channels"])
        heads = int(enc_stage["global_blocks"]["num_mha_heads"])
        checks = [
            (int(module_audio_enc["in_channels"]) == 1, "encoder in_channels must be 1"),
            (int(module_audio_enc.get("groups", 1)) == 1, "encoder groups must be 1"),
            (not module_audio_enc.get("bias", False), "encoder bias is not supported (reference uses bias: false)"),
            (not module_feature_projector.get("bias", False), "projector bias is not supported"),
            (int(module_feature_projector.get("kernel_size", 1)) == 1, "projector kernel_size must be 1"),
            (not module_audio_dec.get("bias", False), "decoder bias is not supported"),
            (int(module_audio_dec["kernel_size"]) == int(module_audio_enc["kernel_size"]), "decoder/encoder kernel mismatch"),
            (int(module_audio_dec["stride"]) == int(module_audio_enc["stride"]), "decoder/encoder stride mismatch"),
            (int(module_audio_dec.get("out_channels", 1)) == 1, "decoder out_channels must be 1"),
            (int(sep["num_stages"]) == int(num_stages), "separator.num_stages != num_stages"),
            (int(sep["spk_split_stage"]["num_spks"]) == int(num_spks), "spk_split_stage.num_spks != num_spks"),
            (int(dec_stage["num_spks"]) == int(num_spks), "dec_stage.num_spks != num_spks"),
            (int(module_output_layer["num_spks"]) == int(num_spks), "output_layer.num_spks != num_spks"),
            (int(module_output_layer["in_channels"]) == int(module_audio_enc["out_channels"]), "output_layer.in_channels != encoder channels"),
            (int(module_output_layer["out_channels"]) == feat, "output_layer.out_channels != F"),
            (int(sep["relative_positional_encoding"]["in_channels"]) == feat, "rel-pos in_channels != F"),
            (int(sep["relative_positional_encoding"]["num_heads"]) == heads, "rel-pos heads != attention heads"),
            (not sep["relative_positional_encoding"].get("embed_v", False), "embed_v is not supported (reference uses false)"),
            (int(dec_stage["global_blocks"]["num_mha_heads"]) == heads, "decoder heads != encoder heads"),
            (int(dec_stage["spk_attention"]["num_mha_heads"]) == heads, "spk-attention heads != encoder heads"),
            (int(sep["simple_fusion"]["out_channels"]) == feat, "simple_fusion.out_channels != F"),
            (int(dec_stage["local_blocks"]["kernel_size"]) == int(enc_stage["local_blocks"]["kernel_size"]), "CLA kernel mismatch enc/dec"),
            (feat % heads == 0, "F must be divisible by heads"),
        ]
        for ok, msg in checks:
            if not ok:
                raise ValueError(f"unsupported model config: {msg}")
        return SepConfig(
            num_stages=int(num_stages), num_spks=int(num_spks),
            enc_channels=int(module_audio_enc["out_channels"]),
            enc_kernel=int(module_audio_enc["kernel_size"]),
            enc_stride=int(module_audio_enc["stride"]),
            feat=feat, heads=heads,
            maxlen=int(sep["relative_positional_encoding"]["maxlen"]),
            cla_kernel=int(enc_stage["local_blocks"]["kernel_size"]),
            down_kernel=int(enc_stage["down_conv_layer"]["samp_kernel_size"]),
            dropout=float(enc_stage["global_blocks"]["dropout_rate"]),
            per_level_split=bool(per_level_split),
        )

    def model_kwargs(self) -> Dict[str, Any]:
        """Inverse of ``from_model_kwargs``: the nested dict ``Model(**kwargs)`` expects."""
        F, H, S, p = self.feat, self.heads, self.num_spks, self.dropout
        g = {"in_channels": F, "num_mha_heads": H, "dropout_rate": p}
        l = {"in_channels": F, "kernel_size": self.cla_kernel, "dropout_rate": p}
        return {
            "num_stages": self.num_stages, "num_spks": S,
            "module_audio_enc": {"in_channels": 1, "out_channels": self.enc_channels,
                                 "kernel_size": self.enc_kernel, "stride": self.enc_stride,
                                 "groups": 1, "bias": False},
            "module_feature_projector": {"num_channels": self.enc_channels, "in_channels": self.enc_channels,
                                         "out_channels": F, "kernel_size": 1, "bias": False},
            "module_separator": {
                "num_stages": self.num_stages,
                "relative_positional_encoding": {"in_channels": F, "num_heads": H,
                                                 "maxlen": self.maxlen, "embed_v": False},
                "enc_stage": {"global_blocks": dict(g), "local_blocks": dict(l),
                              "down_conv_layer": {"in_channels": F, "samp_kernel_size": self.down_kernel}},
                "spk_split_stage": {"in_channels": F, "num_spks": S},
                "simple_fusion": {"out_channels": F},
                "dec_stage": {"num_spks": S, "global_blocks": dict(g), "local_blocks": dict(l),
                              "spk_attention": dict(g)},
            },
            "module_output_layer": {"in_channels": self.enc_channels, "out_channels": F, "num_spks": S},
            "module_audio_dec": {"in_channels": self.enc_channels, "out_channels": 1,
                                 "kernel_size": self.enc_kernel, "stride": self.enc_stride, "bias": False},
        }


# Named variants.  The four reference model directories differ only in F / dropout and, for WHAM, in
# keeping one speaker-split stage per U-Net level.
VARIANTS: Dict[str, SepConfig] = {
    "SepReformer_Base_WSJ0": SepConfig(),
    "SepReformer_Large_DM_WSJ0": SepConfig(feat=256, dropout=0.1),
    "SepReformer_Large_DM_WHAMR": SepConfig(feat=256, dropout=0.1),
    "SepReformer_Large_DM_WHAM": SepConfig(feat=256, dropout=0.1, per_level_split=True),
    # small configuration used by the parity tests (not a reference variant)
    "tiny": SepConfig(num_stages=2, enc_channels=64, feat=64, heads=4, maxlen=40),
    # the same with THREE speakers: the reference is generic in num_spks (modules/module.py:111-118, network.py:241-247)
    "tiny3": SepConfig(num_stages=2, num_spks=3, enc_channels=64, feat=64, heads=4, maxlen=40),
}


def load_model_kwargs(yaml_path: str) -> Dict[str, Any]:
    """Read ``config.model`` from a reference-style ``configs.yaml`` (reference utils/util_system.py:11-29)."""
    with open(yaml_path, "r") as f:
        doc = yaml.full_load(f)
    return doc["config"]["model"]


def variant_yaml(name: str) -> str:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(here, "models", name, "configs.yaml")
