"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (PyTorch fp32 aten) restatement of the reference SepReformer forward pass
(``/root/reference/models/SepReformer_Base_WSJ0/{model.py,modules/module.py,modules/network.py}``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file; ``sepreformer_amd`` never does (its forward raises when the HIP library is missing).

It is a functional walk over a flat ``state_dict`` (reference key names) - no module classes - but it
keeps the reference's *op sequence and tensor layouts* (including every ``permute().contiguous()``),
so its wall-clock on a host CPU is a fair stand-in for the reference's own CPU forward
(``cpu_baseline.kind = "port"``).

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned by *running the imported reference in the build container*: ``tests/golden/make_golden.py``
checks every function below against the corresponding reference module (max |diff| <= 2e-6 relative)
and writes the input/output vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks
the oracle against those committed vectors wherever it runs.

Every function cites the reference lines it restates (paths relative to
``/root/reference/models/SepReformer_Base_WSJ0/``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as TF

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------------
# leaf helpers
# --------------------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return TF.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    w = sd[p + ".weight"]
    return TF.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], 1e-5)


# train-mode switch (oracle/train_oracle.py): BatchNorm1d then uses batch statistics over (batch, frames) and updates the
# running statistics in place with momentum 0.1, exactly what the reference modules do under model.train()
BN_TRAINING = False


def _bn_eval(sd: SD, p: str, x: Tensor) -> Tensor:
    # BatchNorm1d: eval mode = running statistics; eps 1e-5, momentum 0.1 (torch defaults; network.py:167, module.py:69)
    return TF.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                         sd[p + ".weight"], sd[p + ".bias"], BN_TRAINING, 0.1, 1e-5)


# --------------------------------------------------------------------------------------------------
# modules/network.py
# --------------------------------------------------------------------------------------------------
def gcfn(sd: SD, p: str, x: Tensor) -> Tensor:
    """GCFN.forward, network.py:60-66.  x: [b, T, F] -> [b, T, F]."""
    y = _lin(sd, p + ".net1.1", _ln(sd, p + ".net1.0", x))              # :61  LN -> Linear F->6F
    y = y.permute(0, 2, 1).contiguous()                                 # :62
    c = y.shape[1]
    y = TF.conv1d(y, sd[p + ".depthwise.weight"], sd[p + ".depthwise.bias"], padding=1, groups=c)  # :63
    y = y.permute(0, 2, 1).contiguous()                                 # :64
    y = _lin(sd, p + ".net2.2", TF.glu(y, dim=-1))                      # :65  GLU -> Linear 3F->F
    return x + y * sd[p + ".Layer_scale.layer_scale"]                   # :66


def mha(sd: SD, p: str, x: Tensor, pos_k: Optional[Tensor], heads: int) -> Tensor:
    """MultiHeadAttention.forward (mask is always None), network.py:90-124.  x: [n, t, F]."""
    n, t, F = x.shape
    dk = F // heads
    x = _ln(sd, p + ".layer_norm", x)                                   # :99
    q = _lin(sd, p + ".linear_q", x).view(n, -1, heads, dk).transpose(1, 2)   # :100,103
    k = _lin(sd, p + ".linear_k", x).view(n, -1, heads, dk).transpose(1, 2)   # :101,104
    v = _lin(sd, p + ".linear_v", x).view(n, -1, heads, dk).transpose(1, 2)   # :102,105
    A = torch.matmul(q, k.transpose(-2, -1))                            # :106
    if pos_k is not None:
        rq = q.contiguous().view(n * heads, -1, dk).transpose(0, 1)     # :107
        Bm = torch.matmul(rq, pos_k.transpose(-2, -1))                  # :109  [t, n*h, t]
        Bm = Bm.transpose(0, 1).view(n, heads, pos_k.size(0), pos_k.size(1))  # :110
        scores = (A + Bm) / math.sqrt(dk)                               # :111
    else:
        scores = A / math.sqrt(dk)                                      # :113
    attn = torch.softmax(scores, dim=-1)                                # :120
    o = torch.matmul(attn, v)                                           # :122
    o = o.transpose(1, 2).contiguous().view(n, -1, heads * dk)          # :123
    return _lin(sd, p + ".linear_out", o) * sd[p + ".Layer_scale.layer_scale"]  # :124


def ega(sd: SD, p: str, x: Tensor, pos_k: Tensor, heads: int) -> Tensor:
    """EGA.forward, network.py:138-155.  x: [b, F, T] -> [b, T, F]."""
    down_len = pos_k.shape[0]                                           # :145
    x_down = TF.adaptive_avg_pool1d(x, down_len)                        # :146
    x = x.permute(0, 2, 1)                                              # :147
    x_down = x_down.permute(0, 2, 1)                                    # :148
    x_down = mha(sd, p + ".block.self_attn", x_down, pos_k, heads)      # :149
    x_down = x_down.permute(0, 2, 1)                                    # :150
    x_downup = TF.interpolate(x_down, size=x.shape[1], mode="nearest")  # :151 (F.upsample default)
    x_downup = x_downup.permute(0, 2, 1)                                # :152
    gate = torch.sigmoid(_lin(sd, p + ".block.linear.1", _ln(sd, p + ".block.linear.0", x)))  # :132-135
    return x + gate * x_downup                                          # :153


def cla(sd: SD, p: str, x: Tensor) -> Tensor:
    """CLA.forward (eval-mode BatchNorm), network.py:174-187.  x: [b, T, F]."""
    y = _ln(sd, p + ".layer_norm", x)                                   # :175
    y = TF.glu(_lin(sd, p + ".linear1", y), dim=-1)                     # :176-177
    y = y.permute(0, 2, 1)                                              # :178
    c = y.shape[1]
    y = TF.conv1d(y, sd[p + ".dw_conv_1d.weight"], sd[p + ".dw_conv_1d.bias"], padding="same", groups=c)  # :179
    y = y.permute(0, 2, 1)                                              # :180
    y = _lin(sd, p + ".linear2", y)                                     # :181
    y = y.permute(0, 2, 1)                                              # :182
    y = _bn_eval(sd, p + ".BN", y)                                      # :183
    y = y.permute(0, 2, 1)                                              # :184
    y = _lin(sd, p + ".linear3.1", TF.gelu(y))                          # :185 (GELU exact, Linear)
    return x + y * sd[p + ".Layer_scale.layer_scale"]                   # :187


def global_block(sd: SD, p: str, x: Tensor, pos_k: Tensor, heads: int) -> Tensor:
    """GlobalBlock.forward, network.py:198-209.  [b, F, T] -> [b, F, T]."""
    x = ega(sd, p + ".block.ega", x, pos_k, heads)                      # :205
    x = gcfn(sd, p + ".block.gcfn", x)                                  # :206
    return x.permute(0, 2, 1)                                           # :207


def local_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """LocalBlock.forward, network.py:220-224.  [b, T, F] -> [b, T, F]."""
    return gcfn(sd, p + ".block.gcfn", cla(sd, p + ".block.cla", x))    # :221-222


def spk_attention(sd: SD, p: str, x: Tensor, num_spk: int, heads: int) -> Tensor:
    """SpkAttention.forward, network.py:233-252.  [B*S, F, T] -> [B*S, F, T]."""
    B, F, T = x.shape                                                   # :240
    x = x.view(B // num_spk, num_spk, F, T).contiguous()                # :241
    x = x.permute(0, 3, 1, 2).contiguous()                              # :242
    x = x.view(-1, num_spk, F).contiguous()                             # :243
    x = x + mha(sd, p + ".self_attn", x, None, heads)                   # :244
    x = x.view(B // num_spk, T, num_spk, F).contiguous()                # :245
    x = x.permute(0, 2, 3, 1).contiguous()                              # :246
    x = x.view(B, F, T).contiguous()                                    # :247
    x = x.permute(0, 2, 1)                                              # :248
    x = gcfn(sd, p + ".feed_forward", x)                                # :249
    return x.permute(0, 2, 1)                                           # :250


# --------------------------------------------------------------------------------------------------
# modules/module.py
# --------------------------------------------------------------------------------------------------
def audio_encoder(sd: SD, x: Tensor, stride: int) -> Tensor:
    """AudioEncoder.forward, module.py:19-23.  [B, T] -> [B, N, L]."""
    x = x.unsqueeze(0) if x.dim() == 1 else x.unsqueeze(1)              # :20
    return TF.gelu(TF.conv1d(x, sd["audio_encoder.conv1d.weight"], None, stride=stride))  # :21-22


def feature_projector(sd: SD, x: Tensor) -> Tensor:
    """FeatureProjector.forward, module.py:32-35.  GroupNorm(1, N, eps=1e-8) then 1x1 conv."""
    w = sd["feature_projector.norm.weight"]
    x = TF.group_norm(x, 1, w, sd["feature_projector.norm.bias"], 1e-8)  # :33
    return TF.conv1d(x, sd["feature_projector.conv1d.weight"], None)     # :34


def pad_signal(x: Tensor, num_stages: int) -> Tensor:
    """Separator.pad_signal, module.py:220-234 (3-D input case)."""
    if x.dim() not in (2, 3):
        raise RuntimeError("Input can only be 2 or 3 dimensional.")      # :223
    if x.dim() == 2:
        x = x.unsqueeze(1)                                              # :224
    L = 2 ** num_stages                                                 # :225
    nframe = x.size(2)
    rest = 0 if nframe % L == 0 else (nframe // L + 1) * L - nframe     # :229-230
    if rest > 0:
        x = torch.cat([x, x.new_zeros(x.size(0), x.size(1), rest)], dim=-1)  # :231-233
    return x


def rel_pos_k(sd: SD, t: int, maxlen: int) -> Tensor:
    """Index build module.py:196-197 + RelativePositionalEncoding.forward module.py:52-57 -> [t, t, dk]."""
    pos = torch.arange(0, t).long()
    pos = pos[:, None] - pos[None, :]                                   # :197
    pos = pos.clamp(-maxlen, maxlen - 1) + maxlen                       # :53-54
    return TF.embedding(pos, sd["separator.pos_emb.pe_k.weight"])       # :55


def down_conv(sd: SD, p: str, x: Tensor) -> Tensor:
    """DownConvLayer.forward (eval BN), module.py:72-78.  [b, T, F] -> [b, T/2, F]."""
    x = x.permute(0, 2, 1)                                              # :73
    w = sd[p + ".down_conv.weight"]
    k = w.shape[-1]
    x = TF.conv1d(x, w, sd[p + ".down_conv.bias"], stride=2, padding=(k - 1) // 2, groups=w.shape[0])  # :74
    x = TF.gelu(_bn_eval(sd, p + ".BN", x))                             # :75-76
    return x.permute(0, 2, 1)                                           # :77


def enc_stage(sd: SD, p: str, x: Tensor, pos_k: Tensor, heads: int, has_down: bool) -> Tuple[Tensor, Tensor]:
    """SepEncStage.forward, module.py:88-108.  x: [B, F, T]."""
    for i in (1, 2):
        x = global_block(sd, f"{p}.g_block_{i}", x, pos_k, heads)        # :92,97
        x = x.permute(0, 2, 1).contiguous()                             # :93,98
        x = local_block(sd, f"{p}.l_block_{i}", x)                      # :94,99
        x = x.permute(0, 2, 1).contiguous()                             # :95,100
    skip = x                                                            # :102
    if has_down:
        x = x.permute(0, 2, 1).contiguous()                             # :104
        x = down_conv(sd, p + ".downconv", x)                           # :105
        x = x.permute(0, 2, 1).contiguous()                             # :106
    return x, skip


def spk_split(sd: SD, p: str, x: Tensor, num_spks: int) -> Tensor:
    """SpkSplitStage.forward, module.py:120-125.  [B, F, T] -> [B*S, F, T]."""
    x = TF.conv1d(x, sd[p + ".linear.0.weight"], sd[p + ".linear.0.bias"])   # :114
    x = TF.glu(x, dim=-2)                                               # :115
    x = TF.conv1d(x, sd[p + ".linear.2.weight"], sd[p + ".linear.2.bias"])   # :116
    B, _, T = x.shape
    x = x.view(B * num_spks, -1, T).contiguous()                        # :123
    w = sd[p + ".norm.weight"]
    return TF.group_norm(x, 1, w, sd[p + ".norm.bias"], 1e-8)           # :124


def dec_stage(sd: SD, p: str, x: Tensor, pos_k: Tensor, heads: int, num_spk: int) -> Tensor:
    """SepDecStage.forward, module.py:145-170.  x: [B*S, F, T]."""
    for i in (1, 2, 3):
        x = global_block(sd, f"{p}.g_block_{i}", x, pos_k, heads)
        x = x.permute(0, 2, 1).contiguous()
        x = local_block(sd, f"{p}.l_block_{i}", x)
        x = x.permute(0, 2, 1).contiguous()
        x = spk_attention(sd, f"{p}.spk_attn_{i}", x, num_spk, heads)
    return x


def separator(sd: SD, cfg, x: Tensor, taps: Optional[dict] = None) -> Tuple[Tensor, List[Tensor]]:
    """Separator.forward, module.py:190-218.  x: [B, F, L] -> ([B*S, F, L_pad], R stage tensors)."""
    R, H, S = cfg.num_stages, cfg.heads, cfg.num_spks
    x = pad_signal(x, R)                                                # :193
    len_x = x.shape[-1]
    pos_k = rel_pos_k(sd, len_x // 2 ** R, cfg.maxlen)                  # :196-198
    split_name = (lambda i: f"separator.spk_split_blocks.{i}") if cfg.per_level_split else \
                 (lambda i: "separator.spk_split_block")
    skip = []
    for i in range(R):                                                  # :200-203
        x, sk = enc_stage(sd, f"separator.enc_stages.{i}", x, pos_k, H, True)
        if taps is not None:
            taps[f"enc{i}.skip_pre_split"] = sk
        skip.append(spk_split(sd, split_name(i), sk, S))
    x, _ = enc_stage(sd, "separator.bottleneck_G", x, pos_k, H, False)  # :204
    if taps is not None:
        taps["bottleneck"] = x
    x = spk_split(sd, split_name(R), x, S)                              # :205
    outs = []
    for i in range(R):                                                  # :209-215
        outs.append(x)
        j = R - (i + 1)
        x = TF.interpolate(x, size=skip[j].shape[-1], mode="nearest")   # :212
        x = torch.cat([x, skip[j]], dim=1)                              # :213
        x = TF.conv1d(x, sd[f"separator.simple_fusion.{i}.weight"], sd[f"separator.simple_fusion.{i}.bias"])  # :214
        x = dec_stage(sd, f"separator.dec_stages.{i}", x, pos_k, H, S)  # :215
        if taps is not None:
            taps[f"dec{i}"] = x
    return x, outs


RELU_MASKS = None   # test hook, see output_layer; None = the reference's ReLU


def output_layer(sd: SD, p: str, x: Tensor, enc: Tensor, num_spks: int, masking: bool) -> Tensor:
    """OutputLayer.forward (+Masking with ReLU, concat_opt=None), module.py:249-265, network.py:34-43.
    x: [B*S, F, L_pad], enc: [B, N, L] -> [S, B, N, L]."""
    x = x[..., : enc.shape[-1]]                                         # :250
    x = x.permute(0, 2, 1)                                              # :251
    x = _lin(sd, p + ".end_conv1x1.2", TF.glu(_lin(sd, p + ".end_conv1x1.0", x), dim=-1))  # :252
    x = x.permute(0, 2, 1)                                              # :253
    BS, N, L = x.shape
    B = BS // num_spks
    if masking:
        e = enc.expand(num_spks, B, N, L).transpose(0, 1).contiguous().view(B * num_spks, N, L)  # :258-259
        if RELU_MASKS is not None:
            # TEST HOOK (tests/test_train_gpu.py, frozen-gate gradient check): the next 0/1 mask [B*S, N, L] of this iterator
            # replaces the ReLU's own gate, which makes the loss differentiable in the forward activations
            x = x * next(RELU_MASKS) * e
        else:
            x = torch.relu(x) * e                                       # :260, network.py:41
    return x.view(B, num_spks, N, L).transpose(0, 1)                    # :262-264


def audio_decoder(w: Tensor, x: Tensor, stride: int) -> Tensor:
    """AudioDecoder.forward, module.py:278-283.  [B, N, L] -> [B, T]."""
    if x.dim() not in (2, 3):
        raise RuntimeError("AudioDecoder accept 3/4D tensor as input")   # :280
    y = TF.conv_transpose1d(x if x.dim() == 3 else x.unsqueeze(1), w, None, stride=stride)  # :281
    return torch.squeeze(y, dim=1) if torch.squeeze(y).dim() == 1 else torch.squeeze(y)     # :282


# --------------------------------------------------------------------------------------------------
# model.py
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def model_forward(sd: SD, cfg, x: Tensor, taps: Optional[dict] = None):
    """Model.forward, model.py:38-54.  x: [B, T] fp32 -> (list[S] of [B, T], list[R] of list[S] of [B, T])."""
    S, R = cfg.num_spks, cfg.num_stages
    enc = audio_encoder(sd, x, cfg.enc_stride)                          # :39
    feat = feature_projector(sd, enc)                                   # :40
    if taps is not None:
        taps["enc"] = enc
        taps["proj"] = feat
    last, stage_outs = separator(sd, cfg, feat, taps)                   # :41
    o = output_layer(sd, "out_layer", last, enc, S, masking=False)      # :42
    audio = [audio_decoder(sd["audio_decoder.weight"], o[s], cfg.enc_stride) for s in range(S)]  # :43-44
    audio_aux = []
    for i, so in enumerate(stage_outs):                                 # :48-52
        up = TF.interpolate(so, size=enc.shape[-1], mode="nearest")     # :49
        oa = output_layer(sd, f"out_layer_bn.{i}", up, enc, S, masking=True)
        audio_aux.append([audio_decoder(sd[f"decoder_bn.{i}.weight"], oa[s], cfg.enc_stride)[..., : x.shape[-1]]
                          for s in range(S)])                           # :51
    return audio, audio_aux


# --------------------------------------------------------------------------------------------------
# metrics used by the parity gates (reference utils/implements/criterions.py:191-217 semantics)
# --------------------------------------------------------------------------------------------------
def si_snr_db(est: Tensor, ref: Tensor, eps: float = 1e-8) -> Tensor:
    """Zero-mean, optimally-scaled SI-SNR in dB along the last dim (criterions.py:200-209)."""
    est = est.double() - est.double().mean(-1, keepdim=True)
    ref = ref.double() - ref.double().mean(-1, keepdim=True)
    proj = (est * ref).sum(-1, keepdim=True) * ref / ((ref * ref).sum(-1, keepdim=True) + eps)
    noise = est - proj
    return 10.0 * torch.log10((proj * proj).sum(-1) / ((noise * noise).sum(-1) + eps) + eps)


def agreement_db(test: Tensor, truth: Tensor) -> float:
    """10 log10(|truth|^2 / |test - truth|^2) over the whole tensor (plain SNR of the deviation)."""
    t = truth.double()
    d = test.double() - t
    return float(10.0 * torch.log10((t * t).sum() / ((d * d).sum() + 1e-300)))


def pit_si_snr_db(ests: List[Tensor], srcs: List[Tensor]) -> Tensor:
    """Best-permutation mean SI-SNR per utterance for 2 speakers (criterions.py:211-216, no clamp)."""
    assert len(ests) == 2 and len(srcs) == 2
    a = (si_snr_db(ests[0], srcs[0]) + si_snr_db(ests[1], srcs[1])) / 2
    b = (si_snr_db(ests[0], srcs[1]) + si_snr_db(ests[1], srcs[0])) / 2
    return torch.maximum(a, b)
