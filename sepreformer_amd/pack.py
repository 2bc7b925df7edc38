"""Host-side weight packing: reference ``state_dict`` tensors -> the layouts ``include/sepr.h`` documents.

Done once per weight version on the device that owns the parameters (plain torch ops; weights are
"PyTorch-owned storage", the kernels only read them).  What happens here:

* depthwise-conv weights ``[C,1,K]`` become tap-major ``[K,C]`` so a wave reads contiguous channels;
* q/k/v projections are stacked into one ``[3F,F]`` matrix (one launch instead of three);
* eval-mode BatchNorm is folded: CLA's into ``linear2`` (reference ``modules/network.py:181-183``),
  DownConv's into a per-channel scale/shift behind the depthwise conv (``modules/module.py:74-75``);
  folding is done in fp64 and rounded once;
* 1x1 ``Conv1d`` weights ``[O,I,1]`` are viewed ``[O,I]``; ``LayerScale`` ``[1,1,F]`` is viewed ``[F]``;
* encoder / decoder kernels ``[N,1,K]`` become tap-major ``[K,N]``.
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch

from . import lib as L
from .config import SepConfig

BN_EPS = 1e-5      # torch.nn.BatchNorm1d default
GN_EPS = 1e-8      # reference modules/module.py:28,117


PRECISIONS = ("fp32", "bf16x3")          # inference arithmetic (the training path adds "bf16": train_pack.PRECISIONS)


def pack_x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[N,K]`` -> bf16 hi/lo planes in MFMA fragment order ``[N/16][K/32][plane][lane][8]``.

    ``hi = bf16(w)`` (round to nearest even), ``lo = bf16(w - hi)``.  A wave of the bf16x3 core reads the
    64 x 16-byte fragment of one (16-column tile, 32-deep K step, plane) as one contiguous 1 KiB block:
    lane ``g*16 + i`` holds ``w[16*tile + i][32*step + 8*g : 32*step + 8*g + 8]``
    (sepreformer_amd/csrc/sepr_gemm_x3.h)."""
    N, K = w.shape
    if N % 16 or K % 32:
        raise ValueError(f"bf16x3 packing needs N % 16 == 0 and K % 32 == 0, got {N}x{K}")
    w = w.detach().to(torch.float32)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)

    def frag(p):
        return p.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4)      # [tile, step, g, i, 8]

    return torch.stack([frag(hi), frag(lo)], dim=2).contiguous()             # [tile, step, plane, g, i, 8]


class Packed:
    """Packed device tensors + the ctypes structs pointing at them (tensors are kept alive here)."""

    def __init__(self, precision: str = "fp32"):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.precision = precision
        self.fuse_gcfn = os.environ.get("SEPR_FUSE_GCFN", "1") != "0"     # A/B switch for the fused GCFN kernel
        self.fuse_spk = os.environ.get("SEPR_FUSE_SPK", "1") != "0"       # A/B switch for the fused speaker attention
        self.fuse_cla = os.environ.get("SEPR_FUSE_CLA", "1") != "0"       # A/B switch for the fused CLA head / tail
        self.fuse_gate = os.environ.get("SEPR_FUSE_GATE", "1") != "0"     # A/B switch for the fused EGA gate
        self.fuse_mlp = os.environ.get("SEPR_FUSE_MLP", "1") != "0"       # A/B switch for the fused SpkSplit / OutputLayer GLU-MLP
        self.keep: List[torch.Tensor] = []

    def t(self, x: torch.Tensor) -> int:
        x = x.detach().to(torch.float32).contiguous()
        self.keep.append(x)
        return x.data_ptr()

    def x3(self, w: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor = None, beta: torch.Tensor = None) -> L.X3W:
        """bf16x3 form of ``y = LN_affine(x) . w^T + bias`` (NULL struct in fp32 mode).  LayerNorm's affine
        is folded in fp64:  (x*g + b) . w^T = x . (w*g)^T + w . b."""
        if self.precision != "bf16x3":
            return L.X3W()
        w64, b64 = w.detach().double(), bias.detach().double()
        if gamma is not None:
            b64 = b64 + w64 @ beta.detach().double()
            w64 = w64 * gamma.detach().double()[None, :]
        wp = pack_x3(w64.float())
        self.keep.append(wp)
        return L.X3W(wp=wp.data_ptr(), bias=self.t(b64.float()))

    def glumlp_fused(self, w1, b1, w2) -> dict:
        """Fused Linear -> GLU -> Linear (SpkSplit, OutputLayer) for F = 128 in bf16x3 mode; ``{}`` otherwise."""
        F, H, N = w1.shape[1], w1.shape[0] // 2, w2.shape[0]
        if (self.precision != "bf16x3" or F not in (128, 256) or H % 32 or N % F or not self.fuse_mlp
                or (F == 256 and os.environ.get("SEPR_FUSE_MLP256", "0") != "1")):      # (F = 256: built, measured no gain - 426 vs 426 utt/s; off)
            return {}
        w1p, w2p = pack_glumlp_fused(w1, b1, w2)
        self.keep += [w1p, w2p]
        return {"fused_w1p": w1p.data_ptr(), "fused_w2p": w2p.data_ptr()}

    def glumlp_fold(self, w1, w2, b2, dec_weight) -> dict:
        """OutputLayer's second projection with the AudioDecoder folded in (heads without a mask; bf16x3, F = 128, k = 16)."""
        F, H = w1.shape[1], w1.shape[0] // 2
        if (self.precision != "bf16x3" or F not in (128, 256) or H % 32 or w2.shape[0] % F or not self.fuse_mlp or dec_weight.shape[2] != 16
                or (F == 256 and os.environ.get("SEPR_FUSE_MLP256", "0") != "1")
                or os.environ.get("SEPR_FOLD_HEAD", "1") == "0"):
            return {}
        w2p, bf = pack_glumlp_fold(w2, b2, dec_weight)
        self.keep += [w2p, bf]
        return {"fold_w2p": w2p.data_ptr(), "fold_b": bf.data_ptr()}

    def gcfn_fused(self, sd, p: str) -> dict:
        """Fused-GCFN weight forms (bf16x3 mode, F in {64, 128}); empty dict otherwise."""
        F = sd[p + ".net1.0.weight"].shape[0]
        if self.precision != "bf16x3" or F not in (64, 128, 256) or not self.fuse_gcfn or (F == 256 and os.environ.get("SEPR_FUSE_GCFN256", "1") == "0"):
            return {}
        w1p, w2p = pack_gcfn_fused(sd[p + ".net1.1.weight"], sd[p + ".net1.1.bias"], sd[p + ".net1.0.weight"],
                                   sd[p + ".net1.0.bias"], sd[p + ".net2.2.weight"], sd[p + ".depthwise.weight"],
                                   sd[p + ".depthwise.bias"])
        self.keep += [w1p, w2p]
        return {"fused_w1p": w1p.data_ptr(), "fused_w2p": w2p.data_ptr()}


    def cla_fused(self, sd, p: str, w2: torch.Tensor, b2: torch.Tensor) -> dict:
        """Fused CLA head / tail weight forms (bf16x3, F = 128); ``w2`` / ``b2`` = linear2 with BatchNorm folded."""
        F = sd[p + ".layer_norm.weight"].shape[0]
        if self.precision != "bf16x3" or F not in (128, 256) or not self.fuse_cla or (F == 256 and os.environ.get("SEPR_FUSE_CLA256", "1") == "0"):
            return {}
        w1p, w2p, w3p = pack_cla_fused(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], sd[p + ".layer_norm.weight"],
                                       sd[p + ".layer_norm.bias"], w2, b2, sd[p + ".linear3.1.weight"])
        self.keep += [w1p, w2p, w3p]
        return {"fused_w1p": w1p.data_ptr(), "fused_w2p": w2p.data_ptr(), "fused_w3p": w3p.data_ptr()}

    def gate_fused(self, sd, p: str) -> dict:
        """Fused EGA-gate weight form (bf16x3, F = 128)."""
        F = sd[p + ".block.linear.0.weight"].shape[0]
        if self.precision != "bf16x3" or F not in (128, 256) or not self.fuse_gate or (F == 256 and os.environ.get("SEPR_FUSE_GATE256", "1") == "0"):
            return {}
        wp = pack_gate_fused(sd[p + ".block.linear.1.weight"], sd[p + ".block.linear.1.bias"],
                             sd[p + ".block.linear.0.weight"], sd[p + ".block.linear.0.bias"])
        self.keep.append(wp)
        return {"fused_gate_p": wp.data_ptr()}

    def qkv_fused(self, sd, p: str) -> dict:
        """EGA attention in-projection ``[3F, F]`` behind its LayerNorm in the gate's chunk form (bf16x3, F = 128): pooling + LayerNorm +
        q / k / v in one launch (``launch_ega_qkv``).  ``SEPR_FUSE_QKV=0`` keeps pool_stats + the generic projection."""
        F = sd[p + ".layer_norm.weight"].shape[0]
        if self.precision != "bf16x3" or F != 128 or not self.fuse_gate or os.environ.get("SEPR_FUSE_QKV", "1") == "0":
            return {}
        wqkv = torch.cat([sd[f"{p}.linear_{n}.weight"] for n in "qkv"], dim=0)
        bqkv = torch.cat([sd[f"{p}.linear_{n}.bias"] for n in "qkv"], dim=0)
        wp = pack_gate_fused(wqkv, bqkv, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
        wo = _split_frag(sd[p + ".linear_out.weight"].detach().float()).contiguous()     # [F/16][F/32][2][64][8] bf16: the gate launch's folded
        self.keep += [wp, wo]                                                              # output projection (SEPR_FUSE_OUT=0: separate launch)
        out = {"fused_qkv_p": wp.data_ptr()}
        if os.environ.get("SEPR_FUSE_OUT", "1") != "0":
            out["fused_out_p"] = wo.data_ptr()
        return out

    def spk_fused(self, sd, p: str, heads: int, num_spks: int) -> dict:
        """Fused speaker-attention weight forms (bf16x3, F = 128, 16-channel heads, two speakers); else empty."""
        F = sd[p + ".layer_norm.weight"].shape[0]
        if (self.precision != "bf16x3" or (F, F // heads) not in ((128, 16), (256, 32)) or num_spks != 2 or not self.fuse_spk
                or (F == 256 and os.environ.get("SEPR_FUSE_SPK256", "1") == "0")):
            return {}
        wqkv = torch.cat([sd[f"{p}.linear_{n}.weight"] for n in "qkv"], dim=0)
        bqkv = torch.cat([sd[f"{p}.linear_{n}.bias"] for n in "qkv"], dim=0)
        w1p, w2p = pack_spk_fused(wqkv, bqkv, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"],
                                  sd[p + ".linear_out.weight"])
        self.keep += [w1p, w2p]
        return {"fused_qkv_p": w1p.data_ptr(), "fused_out_p": w2p.data_ptr()}


def _split_frag(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[R*16, K]`` (K % 32 == 0) -> ``[R][K/32][plane][64][8]`` bf16 fragments (see ``pack_x3``)."""
    R16, K = w.shape
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)

    def frag(p):
        return p.view(R16 // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(R16 // 16, K // 32, 64, 8)

    return torch.stack([frag(hi), frag(lo)], dim=2)                          # [R, K/32, plane, 64, 8]


def pack_gcfn_fused(w1: torch.Tensor, b1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w2: torch.Tensor,
                    dw_w: torch.Tensor, dw_b: torch.Tensor):
    """Weights of the fully fused GCFN kernel (sepreformer_amd/csrc/sepr_gcfn_fused.hip).

    ``w1`` ``[6F,F]`` (value rows then gate rows), ``w2`` ``[F,3F]``, ``dw_w`` ``[6F,1,3]``, ``dw_b`` ``[6F]``.
    Returns two byte tensors:

    * ``w1p``: per 32-channel hidden chunk, the up-projection fragments ``[4][F/32][2][64][8]`` bf16 (value tiles 0,1
      then their gate tiles; LayerNorm gamma folded) followed by a 4 KB fp32 constants block
      ``[2 tile pairs][b1v b1g wv0 wv1 wv2 wg0 wg1 wg2 cbv cbg][16 channels]`` (bias with beta folded, conv taps,
      conv bias; the gate's taps and conv bias multiplied by -log2(e)), zero padded;
    * ``w2p`` ``[3F/32][F/16][2][64][8]`` bf16: the chunk's K slice of the down-projection with the k-slot order the
      kernel's registers provide (lane group g, slot e -> hidden channel ``e<4 ? 4g+e : 16+4g+e-4`` of the chunk) and
      the output rows of each tile pair interleaved (see below)."""
    F = w1.shape[1]
    H3 = 3 * F
    nch = H3 // 32
    dev = w1.device
    w1f = (w1.detach().double() * gamma.detach().double()[None, :]).float()
    b1f = (b1.detach().double() + w1.detach().double() @ beta.detach().double()).float()
    # the gate half of the depthwise conv (taps and bias) carries the -log2(e) of sigmoid(g) = 1 / (1 + exp2(-log2(e) g)):
    # the kernels evaluate the GLU as value * rcp(1 + exp2(gate')) (csrc/sepr_common.h glu_prescaled)
    gscale = torch.ones(2 * H3, dtype=torch.float64, device=dev)
    gscale[H3:] = -1.4426950408889634
    taps = (dw_w.detach().double()[:, 0, :] * gscale[:, None]).float()           # [6F, 3]
    cb = (dw_b.detach().double() * gscale).float()
    chunks = []
    for c in range(nch):
        rows = []
        for t in range(4):
            base = (0 if t < 2 else H3) + 32 * c + 16 * (t & 1)
            rows.append(w1f[base:base + 16])
        frag = _split_frag(torch.cat(rows, 0)).contiguous().view(torch.uint8).reshape(-1)   # [4][KS][2][64][8] bf16
        cst = torch.zeros(1024, dtype=torch.float32, device=dev)
        for j in range(2):
            v = 32 * c + 16 * j
            g_ = H3 + v
            vals = [b1f[v:v + 16], b1f[g_:g_ + 16], taps[v:v + 16, 0], taps[v:v + 16, 1], taps[v:v + 16, 2],
                    taps[g_:g_ + 16, 0], taps[g_:g_ + 16, 1], taps[g_:g_ + 16, 2], cb[v:v + 16], cb[g_:g_ + 16]]
            cst[j * 160:(j + 1) * 160] = torch.cat(vals)
        chunks.append(torch.cat([frag, cst.view(torch.uint8)]))
    w1p = torch.stack(chunks, 0).contiguous()
    # fragment row 4q+r of tile ft <- output channel 32*(ft//2) + 8q + 4*(ft%2) + r: a lane's accumulators of a tile pair are
    # then 8 consecutive channels of its frame (the kernels' epilogues rely on it)
    ft = torch.arange(F // 16, device=dev)[:, None, None]
    q = torch.arange(4, device=dev)[None, :, None]
    r = torch.arange(4, device=dev)[None, None, :]
    rows = (32 * (ft // 2) + 8 * q + 4 * (ft % 2) + r).reshape(-1)
    w2p = _kslot_frags(w2.detach()[rows], nch)
    return w1p, w2p


def pack_gcfn_fused_batched(w1: torch.Tensor, b1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w2: torch.Tensor,
                            dw_w: torch.Tensor, dw_b: torch.Tensor):
    """``pack_gcfn_fused`` for a STACK of G blocks in a fixed number of torch ops (the training path re-packs all 56 GCFN
    blocks after every optimizer step: ``train_pack.py``).  ``w1 [G,6F,F]``, ``b1 [G,6F]``, ``gamma/beta [G,F]``,
    ``w2 [G,F,3F]``, ``dw_w [G,6F,3]`` (or ``[G,6F,1,3]``), ``dw_b [G,6F]`` -> ``(w1p [G, bytes], w2p [G,3F/32,F/16,2,64,8] bf16)``,
    byte-identical per block to ``pack_gcfn_fused`` (CPU test)."""
    G, _, F = w1.shape
    H3, KS = 3 * F, F // 32
    nch = H3 // 32
    dev = w1.device
    w64 = w1.detach().double()
    w1f = (w64 * gamma.detach().double()[:, None, :]).float()
    b1f = (b1.detach().double() + torch.einsum("gnk,gk->gn", w64, beta.detach().double())).float()
    gscale = torch.ones(2 * H3, dtype=torch.float64, device=dev)
    gscale[H3:] = -1.4426950408889634
    taps = (dw_w.detach().double().reshape(G, 2 * H3, 3) * gscale[None, :, None]).float()      # [G,6F,3]
    cb = (dw_b.detach().double() * gscale[None, :]).float()
    ar = lambda n: torch.arange(n, device=dev)                                                  # noqa: E731
    c, t, i = ar(nch)[:, None, None], ar(4)[None, :, None], ar(16)[None, None, :]
    rows = (torch.where(t < 2, 0, H3) + 32 * c + 16 * (t & 1) + i).reshape(-1)                  # [nch*4*16] chunk, tile, row
    wsel = w1f[:, rows]                                                                         # [G, nch*64, F]
    hi = wsel.to(torch.bfloat16)
    lo = (wsel - hi.to(torch.float32)).to(torch.bfloat16)

    def frag(p):                                                                                # -> [G,nch,tile,KS,g,i,8]
        return p.view(G, nch, 4, 16, KS, 4, 8).permute(0, 1, 2, 4, 5, 3, 6)

    fr = torch.stack([frag(hi), frag(lo)], dim=4).contiguous().view(torch.uint8).reshape(G, nch, -1)
    j = ar(2)[None, :, None]
    v = (32 * ar(nch)[:, None, None] + 16 * j + ar(16)[None, None, :]).reshape(-1)              # [nch*2*16]
    g_ = H3 + v
    vals = torch.stack([b1f[:, v], b1f[:, g_], taps[:, v, 0], taps[:, v, 1], taps[:, v, 2], taps[:, g_, 0], taps[:, g_, 1],
                        taps[:, g_, 2], cb[:, v], cb[:, g_]], dim=1)                            # [G,10,nch*2*16]
    vals = vals.view(G, 10, nch, 2, 16).permute(0, 2, 3, 1, 4).reshape(G, nch, 320)
    cst = torch.zeros(G, nch, 1024, dtype=torch.float32, device=dev)
    cst[:, :, :320] = vals
    w1p = torch.cat([fr, cst.view(torch.uint8).reshape(G, nch, 4096)], dim=2).reshape(G, -1).contiguous()
    ft, q, r = ar(F // 16)[:, None, None], ar(4)[None, :, None], ar(4)[None, None, :]
    orow = (32 * (ft // 2) + 8 * q + 4 * (ft % 2) + r).reshape(-1)
    gk, e = ar(4)[:, None], ar(8)[None, :]
    perm = torch.where(e < 4, 4 * gk + e, 16 + 4 * gk + (e - 4)).reshape(-1)                     # [32] slot (g,e) -> channel
    cols = (32 * ar(nch)[:, None] + perm[None, :]).reshape(-1)
    w2k = w2.detach().float()[:, orow][:, :, cols].view(G, F, nch, 32).permute(0, 2, 1, 3)      # [G,nch,F,32]
    h2 = w2k.to(torch.bfloat16)
    l2 = (w2k - h2.to(torch.float32)).to(torch.bfloat16)

    def frag2(p):                                                                               # -> [G,nch,F/16,g,i,8]
        return p.reshape(G, nch, F // 16, 16, 4, 8).permute(0, 1, 2, 4, 3, 5)

    w2p = torch.stack([frag2(h2), frag2(l2)], dim=3).reshape(G, nch, F // 16, 2, 64, 8).contiguous()
    return w1p, w2p


def pack_glumlp_fused(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor):
    """Weights of the plain GLU-MLP mode of the fused GCFN kernel (``gcfn_fused3_kernel<..., MODE = 1>``):
    ``y = w2 . GLU(w1 x + b1)`` with ``w1`` ``[2H, F]`` (value rows, then gate rows), ``w2`` ``[N, H]``, N a multiple of F (one kernel launch
    writes F output columns: 128 for Base, 256 for the Large variants).

    * ``w1p``: per 32-channel hidden chunk the fragments ``[v0 v1 g0 g1][F/32][2][64][8]`` bf16 + the 4 KB constants block of
      ``pack_gcfn_fused`` with only the biases filled in; the gate rows and their bias carry the -log2(e) of ``glu_prescaled``;
    * ``w2p`` ``[N/F][H/32][F/16][2][64][8]`` bf16: for every block of F output channels the K slices in k-slot order with the
      tile-pair row interleave of ``pack_gcfn_fused`` (one kernel launch per block)."""
    F, H, N = w1.shape[1], w1.shape[0] // 2, w2.shape[0]
    nch = H // 32
    dev = w1.device
    gscale = torch.ones(2 * H, dtype=torch.float64, device=dev)
    gscale[H:] = -1.4426950408889634
    w1f = (w1.detach().double() * gscale[:, None]).float()
    b1f = (b1.detach().double() * gscale).float()
    chunks = []
    for c in range(nch):
        rows = [w1f[(0 if t < 2 else H) + 32 * c + 16 * (t & 1):][:16] for t in range(4)]
        frag = _split_frag(torch.cat(rows, 0)).contiguous().view(torch.uint8).reshape(-1)
        cst = torch.zeros(1024, dtype=torch.float32, device=dev)
        for j in range(2):
            v = 32 * c + 16 * j
            cst[j * 160:j * 160 + 16] = b1f[v:v + 16]
            cst[j * 160 + 16:j * 160 + 32] = b1f[H + v:H + v + 16]
        chunks.append(torch.cat([frag, cst.view(torch.uint8)]))
    w1p = torch.stack(chunks, 0).contiguous()
    ft = torch.arange(F // 16, device=dev)[:, None, None]
    q = torch.arange(4, device=dev)[None, :, None]
    r = torch.arange(4, device=dev)[None, None, :]
    rows = (32 * (ft // 2) + 8 * q + 4 * (ft % 2) + r).reshape(-1)
    w2p = torch.stack([_kslot_frags(w2.detach()[F * h:F * h + F][rows], nch) for h in range(N // F)], 0).contiguous()
    return w1p, w2p


def pack_glumlp_fold(w2: torch.Tensor, b2: torch.Tensor, dec_weight: torch.Tensor):
    """``end_conv1x1.2`` (``w2 [N,H]``, ``b2 [N]``) followed by ``ConvTranspose1d(N -> 1, K, stride)`` (``dec_weight [N,1,K]``, no bias) with
    nothing in between (reference ``modules/module.py:252-256,278-283`` with ``masking=False``, ``model.py:28``) is one linear map from the
    gated tensor to the K taps a frame adds to the waveform: ``W_fold [K,H] = wdec^T . w2``, ``b_fold [K] = wdec^T . b2``, formed in fp64
    and rounded once.  Returns (``[H/32][1][2][64][8]`` bf16 hi/lo k-slot fragments for ``gcfn_fused3_kernel<..., MODE = 2>``, ``b_fold`` fp32)."""
    wd = dec_weight.detach().double()[:, 0, :]                                # [N, K]
    wf = (wd.t() @ w2.detach().double()).float()                              # [K, H]
    bf = (wd.t() @ b2.detach().double()).float().contiguous()
    if wf.shape[0] != 16 or wf.shape[1] % 32:
        raise ValueError(f"folded head needs K = 16 taps and H % 32 == 0, got {tuple(wf.shape)}")
    return _kslot_frags(wf, wf.shape[1] // 32), bf


def _kslot_frags(w2: torch.Tensor, nch: int) -> torch.Tensor:
    """``w2`` ``[F, 32*nch]`` -> ``[nch][F/16][2][64][8]`` bf16: per 32-wide K chunk, the fragments with the k-slot
    order the fused kernels' registers provide (lane group g, slot e -> channel ``e<4 ? 4g+e : 16+4g+e-4``)."""
    dev = w2.device
    g = torch.arange(4, device=dev)[:, None]
    e = torch.arange(8, device=dev)[None, :]
    perm = torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4)).reshape(-1)       # [32] slot (g,e) -> channel
    return torch.stack([_split_frag(w2.detach().float()[:, 32 * c + perm])[:, 0] for c in range(nch)], 0).contiguous()


def _chunk_frags(wm: torch.Tensor, bm: torch.Tensor, bases) -> torch.Tensor:
    """One LDS-DMA chunk of the row-stationary kernels: the bf16 hi/lo fragments of the 16-row tiles starting at
    ``bases`` (``[len(bases)][K/32][2][64][8]``) followed by a 4 KB fp32 block holding their biases ``[len(bases)][16]``."""
    frag = _split_frag(torch.cat([wm[b:b + 16] for b in bases], 0)).contiguous().view(torch.uint8).reshape(-1)
    cst = torch.zeros(1024, dtype=torch.float32, device=wm.device)
    cst[:16 * len(bases)] = torch.cat([bm[b:b + 16] for b in bases])
    return torch.cat([frag, cst.view(torch.uint8)])


def pack_gate_fused(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """EGA gate projection ``[F,F]`` behind a LayerNorm for the fused gate kernel: per 64 output channels four
    16-row tiles (gamma folded into the weights, beta into the bias)."""
    wf = (w.detach().double() * gamma.detach().double()[None, :]).float()
    bf = (b.detach().double() + w.detach().double() @ beta.detach().double()).float()
    return torch.stack([_chunk_frags(wf, bf, [64 * c + 16 * j for j in range(4)]) for c in range(w.shape[0] // 64)], 0).contiguous()


def pack_cla_fused(w1: torch.Tensor, b1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w2: torch.Tensor,
                   b2: torch.Tensor, w3: torch.Tensor):
    """Weights of the fused CLA head / tail kernels (sepreformer_amd/csrc/sepr_cla_fused.hip).

    ``w1`` ``[2F,F]`` (value rows then gate rows, LayerNorm in front), ``w2`` ``[2F,F]`` / ``b2`` with the eval
    BatchNorm already folded, ``w3`` ``[F,2F]``.  Returns three byte tensors:

    * ``w1p``: per 32 output channels the fragments ``[v0 v1 g0 g1][F/32][2][64][8]`` bf16 (gamma folded) + a 4 KB fp32
      constants block ``[v0 v1 g0 g1][16]`` of biases (beta folded), zero padded;
    * ``w2p``: per 64 hidden channels the fragments ``[4 tiles][F/32][2][64][8]`` + 4 KB constants ``[4][16]`` biases;
    * ``w3p`` ``[2F/32][F/16][2][64][8]``: linear3 in k-slot order (``_kslot_frags``)."""
    F = w1.shape[1]
    dev = w1.device
    w1f = (w1.detach().double() * gamma.detach().double()[None, :]).float()
    b1f = (b1.detach().double() + w1.detach().double() @ beta.detach().double()).float()
    w2f, b2f = w2.detach().float(), b2.detach().float()

    def chunk(wm, bm, bases):
        frag = _split_frag(torch.cat([wm[b:b + 16] for b in bases], 0)).contiguous().view(torch.uint8).reshape(-1)
        cst = torch.zeros(1024, dtype=torch.float32, device=dev)
        cst[:16 * len(bases)] = torch.cat([bm[b:b + 16] for b in bases])
        return torch.cat([frag, cst.view(torch.uint8)])

    w1p = torch.stack([chunk(w1f, b1f, [32 * c, 32 * c + 16, F + 32 * c, F + 32 * c + 16]) for c in range(F // 32)], 0)
    w2p = torch.stack([chunk(w2f, b2f, [64 * c + 16 * j for j in range(4)]) for c in range(2 * F // 64)], 0)
    return w1p.contiguous(), w2p.contiguous(), _kslot_frags(w3, 2 * F // 32)


def pack_spk_fused(wqkv: torch.Tensor, bqkv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wo: torch.Tensor):
    """Weights of the fused speaker-attention kernel (sepreformer_amd/csrc/sepr_spk_fused.hip), 16-channel heads.

    ``wqkv`` ``[3F,F]`` (q rows, k rows, v rows), ``wo`` ``[F,F]``.  Returns two byte tensors:

    * ``w1p``: per head pair ``c`` the fragments ``[6][F/32][2][64][8]`` bf16 of the tiles
      ``q(2c) q(2c+1) k(2c) k(2c+1) v(2c) v(2c+1)`` (LayerNorm gamma folded) followed by a 4 KB fp32 constants block
      ``[6 tiles][16 channels]`` of biases (beta folded), zero padded;
    * ``w2p`` ``[F/32][F/16][2][64][8]`` bf16: the pair's K slice of the output projection in k-slot order."""
    F = wqkv.shape[1]
    dev = wqkv.device
    wf = (wqkv.detach().double() * gamma.detach().double()[None, :]).float()
    bf = (bqkv.detach().double() + wqkv.detach().double() @ beta.detach().double()).float()
    chunks = []
    for c in range(F // 32):
        bases = [tt * F + 16 * (2 * c + hh) for tt in range(3) for hh in range(2)]
        frag = _split_frag(torch.cat([wf[b:b + 16] for b in bases], 0)).contiguous().view(torch.uint8).reshape(-1)
        cst = torch.zeros(1024, dtype=torch.float32, device=dev)
        cst[:96] = torch.cat([bf[b:b + 16] for b in bases])
        chunks.append(torch.cat([frag, cst.view(torch.uint8)]))
    return torch.stack(chunks, 0).contiguous(), _kslot_frags(wo, F // 32)


def _tapmajor(w: torch.Tensor) -> torch.Tensor:
    return w[:, 0, :].t().contiguous()          # [C,1,K] -> [K,C]


def pack_gcfn(pk: Packed, sd: Dict[str, torch.Tensor], p: str) -> L.GcfnW:
    return L.GcfnW(
        ln_g=pk.t(sd[p + ".net1.0.weight"]), ln_b=pk.t(sd[p + ".net1.0.bias"]),
        w1=pk.t(sd[p + ".net1.1.weight"]), b1=pk.t(sd[p + ".net1.1.bias"]),
        dw_w=pk.t(_tapmajor(sd[p + ".depthwise.weight"])), dw_b=pk.t(sd[p + ".depthwise.bias"]),
        w2=pk.t(sd[p + ".net2.2.weight"]), b2=pk.t(sd[p + ".net2.2.bias"]),
        ls=pk.t(sd[p + ".Layer_scale.layer_scale"].reshape(-1)),
        x3_up=pk.x3(sd[p + ".net1.1.weight"], sd[p + ".net1.1.bias"], sd[p + ".net1.0.weight"], sd[p + ".net1.0.bias"]),
        x3_down=pk.x3(sd[p + ".net2.2.weight"], sd[p + ".net2.2.bias"]),
        **pk.gcfn_fused(sd, p))


def pack_cla(pk: Packed, sd, p: str) -> L.ClaW:
    s = sd[p + ".BN.weight"].double() / torch.sqrt(sd[p + ".BN.running_var"].double() + BN_EPS)
    w2 = (sd[p + ".linear2.weight"].double() * s[:, None]).float()
    b2 = ((sd[p + ".linear2.bias"].double() - sd[p + ".BN.running_mean"].double()) * s + sd[p + ".BN.bias"].double()).float()
    return L.ClaW(
        ln_g=pk.t(sd[p + ".layer_norm.weight"]), ln_b=pk.t(sd[p + ".layer_norm.bias"]),
        w1=pk.t(sd[p + ".linear1.weight"]), b1=pk.t(sd[p + ".linear1.bias"]),
        dw_w=pk.t(_tapmajor(sd[p + ".dw_conv_1d.weight"])), dw_b=pk.t(sd[p + ".dw_conv_1d.bias"]),
        w2=pk.t(w2), b2=pk.t(b2),
        w3=pk.t(sd[p + ".linear3.1.weight"]), b3=pk.t(sd[p + ".linear3.1.bias"]),
        ls=pk.t(sd[p + ".Layer_scale.layer_scale"].reshape(-1)),
        x3_1=pk.x3(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"]),
        x3_2=pk.x3(w2, b2),
        x3_3=pk.x3(sd[p + ".linear3.1.weight"], sd[p + ".linear3.1.bias"]),
        **pk.cla_fused(sd, p, w2, b2))


def pack_mha(pk: Packed, sd, p: str, spk_fused: dict = None) -> L.MhaW:
    wqkv = torch.cat([sd[f"{p}.linear_{n}.weight"] for n in "qkv"], dim=0)
    bqkv = torch.cat([sd[f"{p}.linear_{n}.bias"] for n in "qkv"], dim=0)
    return L.MhaW(
        ln_g=pk.t(sd[p + ".layer_norm.weight"]), ln_b=pk.t(sd[p + ".layer_norm.bias"]),
        wqkv=pk.t(wqkv), bqkv=pk.t(bqkv),
        wo=pk.t(sd[p + ".linear_out.weight"]), bo=pk.t(sd[p + ".linear_out.bias"]),
        ls=pk.t(sd[p + ".Layer_scale.layer_scale"].reshape(-1)),
        x3_qkv=pk.x3(wqkv, bqkv, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"]),
        x3_out=pk.x3(sd[p + ".linear_out.weight"], sd[p + ".linear_out.bias"]),
        **(spk_fused or {}))


def pack_ega(pk: Packed, sd, p: str, pe_ptr: int, maxlen: int, pe_planes_ptr: int = 0) -> L.EgaW:
    return L.EgaW(
        attn=pack_mha(pk, sd, p + ".block.self_attn"),
        gate_ln_g=pk.t(sd[p + ".block.linear.0.weight"]), gate_ln_b=pk.t(sd[p + ".block.linear.0.bias"]),
        gate_w=pk.t(sd[p + ".block.linear.1.weight"]), gate_b=pk.t(sd[p + ".block.linear.1.bias"]),
        pe_k=pe_ptr, maxlen=maxlen, pe_k_planes=pe_planes_ptr,
        x3_gate=pk.x3(sd[p + ".block.linear.1.weight"], sd[p + ".block.linear.1.bias"],
                      sd[p + ".block.linear.0.weight"], sd[p + ".block.linear.0.bias"]),
        **pk.gate_fused(sd, p), **pk.qkv_fused(sd, p + ".block.self_attn"))


def pack_down(pk: Packed, sd, p: str) -> L.DownW:
    s = sd[p + ".BN.weight"].double() / torch.sqrt(sd[p + ".BN.running_var"].double() + BN_EPS)
    shift = (sd[p + ".down_conv.bias"].double() - sd[p + ".BN.running_mean"].double()) * s + sd[p + ".BN.bias"].double()
    return L.DownW(w=pk.t(_tapmajor(sd[p + ".down_conv.weight"])), scale=pk.t(s.float()), shift=pk.t(shift.float()))


def pack_split(pk: Packed, sd, p: str) -> L.SplitW:
    return L.SplitW(
        w1=pk.t(sd[p + ".linear.0.weight"][:, :, 0]), b1=pk.t(sd[p + ".linear.0.bias"]),
        w2=pk.t(sd[p + ".linear.2.weight"][:, :, 0]), b2=pk.t(sd[p + ".linear.2.bias"]),
        gn_g=pk.t(sd[p + ".norm.weight"]), gn_b=pk.t(sd[p + ".norm.bias"]),
        x3_1=pk.x3(sd[p + ".linear.0.weight"][:, :, 0], sd[p + ".linear.0.bias"]),
        x3_2=pk.x3(sd[p + ".linear.2.weight"][:, :, 0], sd[p + ".linear.2.bias"]),
        **pk.glumlp_fused(sd[p + ".linear.0.weight"][:, :, 0], sd[p + ".linear.0.bias"], sd[p + ".linear.2.weight"][:, :, 0]))


def pack_fuse(pk: Packed, sd, p: str) -> L.FuseW:
    w, b = sd[p + ".weight"][:, :, 0], sd[p + ".bias"]
    return L.FuseW(w=pk.t(w), b=pk.t(b), x3=pk.x3(w, b))


def pack_out(pk: Packed, sd, p: str, dec_weight: torch.Tensor, fold: bool = False) -> L.OutW:
    return L.OutW(
        w1=pk.t(sd[p + ".end_conv1x1.0.weight"]), b1=pk.t(sd[p + ".end_conv1x1.0.bias"]),
        w2=pk.t(sd[p + ".end_conv1x1.2.weight"]), b2=pk.t(sd[p + ".end_conv1x1.2.bias"]),
        wdec=pk.t(_tapmajor(dec_weight)),
        x3_1=pk.x3(sd[p + ".end_conv1x1.0.weight"], sd[p + ".end_conv1x1.0.bias"]),
        x3_2=pk.x3(sd[p + ".end_conv1x1.2.weight"], sd[p + ".end_conv1x1.2.bias"]),
        **pk.glumlp_fused(sd[p + ".end_conv1x1.0.weight"], sd[p + ".end_conv1x1.0.bias"], sd[p + ".end_conv1x1.2.weight"]),
        **(pk.glumlp_fold(sd[p + ".end_conv1x1.0.weight"], sd[p + ".end_conv1x1.2.weight"], sd[p + ".end_conv1x1.2.bias"], dec_weight)
           if fold else {}))


class PackedModel(Packed):
    """All blocks of one model, addressed the way the forward driver walks them."""

    def __init__(self, cfg: SepConfig, sd: Dict[str, torch.Tensor], precision: str = "fp32"):
        super().__init__(precision)
        R = cfg.num_stages
        self.cfg = cfg
        self.enc_w = self.t(_tapmajor(sd["audio_encoder.conv1d.weight"]))
        self.proj_g = self.t(sd["feature_projector.norm.weight"])
        self.proj_b = self.t(sd["feature_projector.norm.bias"])
        self.proj_w = self.t(sd["feature_projector.conv1d.weight"][:, :, 0])
        pe = self.t(sd["separator.pos_emb.pe_k.weight"])
        pe_planes = 0
        if precision == "bf16x3":       # the table as bf16 hi / lo planes, split once here instead of per key tile in the attention kernel
            pe32 = sd["separator.pos_emb.pe_k.weight"].to(torch.float32)
            hi = pe32.to(torch.bfloat16)
            lo = (pe32 - hi.to(torch.float32)).to(torch.bfloat16)
            planes = torch.stack([hi, lo], 0).contiguous()
            self.keep.append(planes)
            pe_planes = planes.data_ptr()

        def glob(p):
            return pack_ega(self, sd, p + ".block.ega", pe, cfg.maxlen, pe_planes), pack_gcfn(self, sd, p + ".block.gcfn")

        def loc(p):
            return pack_cla(self, sd, p + ".block.cla"), pack_gcfn(self, sd, p + ".block.gcfn")

        def enc_stage(p, down):
            st = {"g": [glob(f"{p}.g_block_{i}") for i in (1, 2)], "l": [loc(f"{p}.l_block_{i}") for i in (1, 2)]}
            st["down"] = pack_down(self, sd, p + ".downconv") if down else None
            return st

        self.enc_stages = [enc_stage(f"separator.enc_stages.{i}", True) for i in range(R)]
        self.bottleneck = enc_stage("separator.bottleneck_G", False)
        if cfg.per_level_split:
            self.splits = [pack_split(self, sd, f"separator.spk_split_blocks.{i}") for i in range(R + 1)]
        else:
            one = pack_split(self, sd, "separator.spk_split_block")
            self.splits = [one] * (R + 1)
        self.fuse = [pack_fuse(self, sd, f"separator.simple_fusion.{i}") for i in range(R)]
        self.dec_stages = []
        for i in range(R):
            p = f"separator.dec_stages.{i}"
            self.dec_stages.append({
                "g": [glob(f"{p}.g_block_{j}") for j in (1, 2, 3)],
                "l": [loc(f"{p}.l_block_{j}") for j in (1, 2, 3)],
                "spk": [(pack_mha(self, sd, f"{p}.spk_attn_{j}.self_attn",
                                  self.spk_fused(sd, f"{p}.spk_attn_{j}.self_attn", cfg.heads, cfg.num_spks)),
                         pack_gcfn(self, sd, f"{p}.spk_attn_{j}.feed_forward")) for j in (1, 2, 3)],
            })
        self.out_main = pack_out(self, sd, "out_layer", sd["audio_decoder.weight"], fold=True)   # no mask on the main head (model.py:28)
        self.out_aux = [pack_out(self, sd, f"out_layer_bn.{i}", sd[f"decoder_bn.{i}.weight"]) for i in range(R)]
