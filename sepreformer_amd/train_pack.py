"""Per-step weight forms of the training path (``include/sepr.h``, "Training path").

Every optimizer step changes every weight, so these forms are rebuilt once per training forward.  To keep that cheap the
blocks are packed BY TYPE: the parameters of all 56 GCFN blocks (22 EGA, 22 CLA, ...) are stacked once and every fold /
transpose / bf16 split is one batched torch op over the stack (~200 launches per step instead of ~6000); a block's struct
then just points at its slice.  What is built per projection ``y = norm_affine(x) . W^T + b`` (``sepr_lin``):

* forward form: ``W * gamma`` and ``b + W . beta`` (LayerNorm / GroupNorm affine folded in fp64, like ``pack.py``);
* input-gradient form: the transpose ``(W * gamma)^T``, and for a projection followed by LayerScale ``(ls * W)^T``;
* exact-f32 mode keeps fp32 ``[N,K]`` matrices, bf16x3 mode the ``pack_x3`` fragments (hi/lo bf16 planes); ``bf16`` mode
  (plain bf16 operands, one MFMA per product, fp32 accumulate and fp32 master weights: the training precision BASELINE
  configs[4] names) uses the same fragments with ``sepr_lin.planes = 1`` (the kernels then read the hi plane only);
* GCFN blocks additionally get the fused kernel's weight forms (``pack.pack_gcfn_fused_batched``) when F is 64 or 128: their
  train forward is then one launch that keeps only the LayerNorm statistics (``include/sepr.h`` sepr_gcfn_tw).

The raw parameters ride along for the gradient finishers, and a ``*Grad`` struct per block points into one flat fp32
gradient buffer laid out exactly like the parameters (``GradBuffer``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import os

import torch

from . import lib as L
from .config import SepConfig
from .params import KINDS, param_rows

PRECISIONS = ("fp32", "bf16x3", "bf16")


class GradBuffer:
    """One flat fp32 buffer holding the gradient of every parameter (256-byte aligned slices, parameter layouts)."""

    def __init__(self, cfg: SepConfig, device: torch.device):
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape, kind, _ in param_rows(cfg):
            if KINDS[kind]:
                continue
            n = 1
            for d in shape:
                n *= d
            self.offsets[name] = (off, tuple(shape))
            off += (n + 63) // 64 * 64
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)

    def ptr(self, name: str) -> int:
        return self.flat.data_ptr() + 4 * self.offsets[name][0]

    def view(self, name: str) -> torch.Tensor:
        off, shape = self.offsets[name]
        n = 1
        for d in shape:
            n *= d
        return self.flat[off:off + n].view(shape)


def _pack_x3_batched(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[G,N,K]`` -> ``[G, N/16, K/32, 2, 4, 16, 8]`` bf16 fragments (``pack.pack_x3`` per matrix)."""
    G, N, K = w.shape
    if N % 16 or K % 32:
        raise ValueError(f"bf16x3 packing needs N % 16 == 0 and K % 32 == 0, got {N}x{K}")
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)

    def frag(p):
        return p.view(G, N // 16, 16, K // 32, 4, 8).permute(0, 1, 3, 4, 2, 5)      # [G, tile, step, g, i, 8]

    return torch.stack([frag(hi), frag(lo)], dim=3).contiguous()


class _Stack:
    """A stack of G same-shaped projections in one precision; ``lin(i)`` is block i's ``sepr_lin``."""

    def __init__(self, keep: list, w: torch.Tensor, b: Optional[torch.Tensor], precision: str):
        w = w.to(torch.float32).contiguous()
        self.G, self.N, self.K = w.shape
        self.b = None if b is None else b.to(torch.float32).contiguous()
        self.planes = 1 if precision == "bf16" else 0
        if precision in ("bf16x3", "bf16"):
            self.wp = _pack_x3_batched(w)
            self.w = None
            keep.append(self.wp)
        else:
            self.w = w
            self.wp = None
            keep.append(self.w)
        if self.b is not None:
            keep.append(self.b)

    def lin(self, i: int) -> L.Lin:
        bptr = None if self.b is None else self.b.data_ptr() + 4 * i * self.N
        if self.wp is not None:
            return L.Lin(w=None, wp=self.wp.data_ptr() + 2 * i * 2 * self.N * self.K, b=bptr, planes=self.planes)
        return L.Lin(w=self.w.data_ptr() + 4 * i * self.N * self.K, wp=None, b=bptr, planes=0)


def _fold(w: torch.Tensor, b: Optional[torch.Tensor], g: torch.Tensor, beta: torch.Tensor):
    """``(x g + beta) . w^T + b = x . (w g)^T + (b + w . beta)`` for stacks ``w [G,N,K]``, ``g/beta [G,K]`` (fp64)."""
    w64 = w.double()
    wf = (w64 * g.double()[:, None, :]).float()
    bf = torch.einsum("gnk,gk->gn", w64, beta.double())
    if b is not None:
        bf = bf + b.double()
    return wf, bf.float()


def _t(w: torch.Tensor) -> torch.Tensor:
    return w.transpose(1, 2).contiguous()


class TrainPack:
    """All blocks of one model in training form, addressed the way ``train_engine`` walks them."""

    def __init__(self, cfg: SepConfig, sd: Dict[str, torch.Tensor], grads: GradBuffer, precision: str = "bf16x3",
                 salt: Optional[torch.Tensor] = None):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.cfg, self.precision, self.keep = cfg, precision, []
        self.sd, self.gb = sd, grads
        dev = next(iter(sd.values())).device
        F, S, N = cfg.feat, cfg.num_spks, cfg.enc_channels
        R = cfg.num_stages
        # device word XOR-ed into every dropout seed of the step (include/sepr.h seed_salt): zero in eager mode (the by-value
        # seeds change every step), rewritten before each replay of a captured step (model.py, SEPR_TRAIN_GRAPHS)
        # (a captured step passes its own word in: one created here would be re-zeroed by every replay of the capture)
        self.salt = salt if salt is not None else torch.zeros(1, dtype=torch.int64, device=dev)
        self.zeros = torch.zeros(max(2 * F, N, 8 * F), dtype=torch.float32, device=dev)
        self.ones = torch.ones(max(2 * F, N), dtype=torch.float32, device=dev)

        # ---- enumerate the blocks in the engine's walk order ------------------------------------------------------
        gcfn_p: List[str] = []
        ega_p: List[str] = []
        cla_p: List[str] = []
        spk_p: List[str] = []

        def stage_names(p, n_pairs, with_spk):
            out = {"g": [], "l": [], "spk": []}
            for j in range(1, n_pairs + 1):
                out["g"].append((len(ega_p), len(gcfn_p)))
                ega_p.append(f"{p}.g_block_{j}.block.ega")
                gcfn_p.append(f"{p}.g_block_{j}.block.gcfn")
                out["l"].append((len(cla_p), len(gcfn_p)))
                cla_p.append(f"{p}.l_block_{j}.block.cla")
                gcfn_p.append(f"{p}.l_block_{j}.block.gcfn")
                if with_spk:
                    out["spk"].append((len(spk_p), len(gcfn_p)))
                    spk_p.append(f"{p}.spk_attn_{j}.self_attn")
                    gcfn_p.append(f"{p}.spk_attn_{j}.feed_forward")
            return out

        enc_idx = [stage_names(f"separator.enc_stages.{i}", 2, False) for i in range(R)]
        bott_idx = stage_names("separator.bottleneck_G", 2, False)
        dec_idx = [stage_names(f"separator.dec_stages.{i}", 3, True) for i in range(R)]
        down_p = [f"separator.enc_stages.{i}.downconv" for i in range(R)]
        split_p = ([f"separator.spk_split_blocks.{i}" for i in range(R + 1)] if cfg.per_level_split else ["separator.spk_split_block"])
        fuse_p = [f"separator.simple_fusion.{i}" for i in range(R)]
        out_p = ["out_layer"] + [f"out_layer_bn.{i}" for i in range(R)]
        dec_w = ["audio_decoder.weight"] + [f"decoder_bn.{i}.weight" for i in range(R)]

        # state_dict prefixes in pack order: self.gcfn[i] belongs to block_prefixes["gcfn"][i] (tests address single blocks by name)
        self.block_prefixes = {"gcfn": gcfn_p, "ega": ega_p, "cla": cla_p, "spk": spk_p}

        def st(names, suffix, view=None):
            ts = [sd[n + suffix] for n in names]
            t = torch.stack([x.reshape(view) if view is not None else x for x in ts], 0).to(torch.float32)
            return t

        def raw(t):       # keep a contiguous fp32 stack alive; returns (tensor, per-block stride in bytes)
            t = t.contiguous()
            self.keep.append(t)
            return t

        def at(t, i):
            return t.data_ptr() + 4 * i * (t.numel() // t.shape[0])

        gp, P = grads.ptr, precision
        # ---- GCFN --------------------------------------------------------------------------------------------------
        g_ln_g, g_ln_b = raw(st(gcfn_p, ".net1.0.weight")), raw(st(gcfn_p, ".net1.0.bias"))
        g_w1, g_b1 = raw(st(gcfn_p, ".net1.1.weight")), st(gcfn_p, ".net1.1.bias")
        g_w2, g_b2 = raw(st(gcfn_p, ".net2.2.weight")), raw(st(gcfn_p, ".net2.2.bias"))
        g_ls = raw(st(gcfn_p, ".Layer_scale.layer_scale", view=(F,)))
        g_dw = raw(st(gcfn_p, ".depthwise.weight", view=(6 * F, 3)).transpose(1, 2))          # [G,3,6F] tap-major
        g_db = raw(st(gcfn_p, ".depthwise.bias"))
        w1f, b1f = _fold(g_w1, g_b1, g_ln_g, g_ln_b)
        s_up, s_up_t = _Stack(self.keep, w1f, b1f, P), _Stack(self.keep, _t(w1f), None, P)
        s_dn = _Stack(self.keep, g_w2, g_b2, P)
        s_dn_t = _Stack(self.keep, _t(g_w2 * g_ls[:, :, None]), None, P)
        # fused forward (+ statistics-only context) and recomputing backward: packed-bf16 precisions, F in {64, 128}
        self.fused_gcfn = (P in ("bf16x3", "bf16") and F in (64, 128) and os.environ.get("SEPR_TRAIN_FUSE_GCFN", "1") != "0")
        fw1 = fw2 = None
        if self.fused_gcfn:
            from .pack import pack_gcfn_fused_batched
            fw1, fw2 = pack_gcfn_fused_batched(g_w1, g_b1, g_ln_g, g_ln_b, g_w2, st(gcfn_p, ".depthwise.weight", view=(6 * F, 3)), g_db)
            self.keep += [fw1, fw2]
        self.gcfn = []
        for i, p in enumerate(gcfn_p):
            tw = L.GcfnTW(up=s_up.lin(i), up_t=s_up_t.lin(i), down=s_dn.lin(i), down_t=s_dn_t.lin(i), dw_w=at(g_dw, i), dw_b=at(g_db, i),
                          ls=at(g_ls, i), w1=at(g_w1, i), ln_g=at(g_ln_g, i), ln_b=at(g_ln_b, i), w2=at(g_w2, i), b2=at(g_b2, i),
                          fused_w1p=None if fw1 is None else fw1.data_ptr() + i * fw1.shape[1],
                          fused_w2p=None if fw2 is None else fw2.data_ptr() + i * fw2[0].numel() * 2, seed_salt=self.salt.data_ptr())
            gr = L.GcfnGrad(ln_g=gp(p + ".net1.0.weight"), ln_b=gp(p + ".net1.0.bias"), w1=gp(p + ".net1.1.weight"), b1=gp(p + ".net1.1.bias"),
                            dw_w=gp(p + ".depthwise.weight"), dw_b=gp(p + ".depthwise.bias"), w2=gp(p + ".net2.2.weight"),
                            b2=gp(p + ".net2.2.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
            self.gcfn.append((tw, gr))

        # ---- MHA (EGA's and SpkAttention's) ------------------------------------------------------------------------
        def mha_stack(names):
            ln_g, ln_b = raw(st(names, ".layer_norm.weight")), raw(st(names, ".layer_norm.bias"))
            wqkv = raw(torch.cat([st(names, f".linear_{c}.weight") for c in "qkv"], 1))       # [G,3F,F]
            bqkv = torch.cat([st(names, f".linear_{c}.bias") for c in "qkv"], 1)
            wo, bo = raw(st(names, ".linear_out.weight")), raw(st(names, ".linear_out.bias"))
            ls = raw(st(names, ".Layer_scale.layer_scale", view=(F,)))
            wf, bf = _fold(wqkv, bqkv, ln_g, ln_b)
            s_qkv, s_qkv_t = _Stack(self.keep, wf, bf, P), _Stack(self.keep, _t(wf), None, P)
            s_out, s_out_t = _Stack(self.keep, wo, bo, P), _Stack(self.keep, _t(wo * ls[:, :, None]), None, P)
            res = []
            for i, p in enumerate(names):
                tw = L.MhaTW(qkv=s_qkv.lin(i), qkv_t=s_qkv_t.lin(i), out=s_out.lin(i), out_t=s_out_t.lin(i), ls=at(ls, i),
                             wqkv=at(wqkv, i), ln_g=at(ln_g, i), ln_b=at(ln_b, i), wo=at(wo, i), bo=at(bo, i), seed_salt=self.salt.data_ptr())
                gr = L.MhaGrad(ln_g=gp(p + ".layer_norm.weight"), ln_b=gp(p + ".layer_norm.bias"),
                               wq=gp(p + ".linear_q.weight"), bq=gp(p + ".linear_q.bias"), wk=gp(p + ".linear_k.weight"),
                               bk=gp(p + ".linear_k.bias"), wv=gp(p + ".linear_v.weight"), bv=gp(p + ".linear_v.bias"),
                               wo=gp(p + ".linear_out.weight"), bo=gp(p + ".linear_out.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
                res.append((tw, gr))
            return res

        ega_mha = mha_stack([p + ".block.self_attn" for p in ega_p])
        self.spk = mha_stack(spk_p) if spk_p else []
        # ---- EGA gate ------------------------------------------------------------------------------------------------
        e_ln_g, e_ln_b = raw(st(ega_p, ".block.linear.0.weight")), raw(st(ega_p, ".block.linear.0.bias"))
        e_w, e_b = raw(st(ega_p, ".block.linear.1.weight")), st(ega_p, ".block.linear.1.bias")
        ewf, ebf = _fold(e_w, e_b, e_ln_g, e_ln_b)
        s_gate, s_gate_t = _Stack(self.keep, ewf, ebf, P), _Stack(self.keep, _t(ewf), None, P)
        pe = raw(sd["separator.pos_emb.pe_k.weight"].to(torch.float32)[None])
        self.ega = []
        for i, p in enumerate(ega_p):
            tw = L.EgaTW(attn=ega_mha[i][0], gate=s_gate.lin(i), gate_t=s_gate_t.lin(i), gate_w=at(e_w, i), gate_ln_g=at(e_ln_g, i),
                         gate_ln_b=at(e_ln_b, i), pe_k=pe.data_ptr(), maxlen=cfg.maxlen)
            gr = L.EgaGrad(attn=ega_mha[i][1], gate_ln_g=gp(p + ".block.linear.0.weight"), gate_ln_b=gp(p + ".block.linear.0.bias"),
                           gate_w=gp(p + ".block.linear.1.weight"), gate_b=gp(p + ".block.linear.1.bias"),
                           pe_k=gp("separator.pos_emb.pe_k.weight"))
            self.ega.append((tw, gr))

        # ---- CLA -----------------------------------------------------------------------------------------------------
        K = cfg.cla_kernel
        c_ln_g, c_ln_b = raw(st(cla_p, ".layer_norm.weight")), raw(st(cla_p, ".layer_norm.bias"))
        c_w1, c_b1 = raw(st(cla_p, ".linear1.weight")), st(cla_p, ".linear1.bias")
        c_w2, c_b2 = st(cla_p, ".linear2.weight"), st(cla_p, ".linear2.bias")
        c_w3, c_b3 = raw(st(cla_p, ".linear3.1.weight")), raw(st(cla_p, ".linear3.1.bias"))
        c_ls = raw(st(cla_p, ".Layer_scale.layer_scale", view=(F,)))
        c_dw = st(cla_p, ".dw_conv_1d.weight", view=(F, K))
        c_dwt, c_dwf = raw(c_dw.transpose(1, 2)), raw(c_dw.flip(2).transpose(1, 2))          # [G,K,F] tap-major / reversed taps
        c_db = raw(st(cla_p, ".dw_conv_1d.bias"))
        c_bn_g, c_bn_b = raw(st(cla_p, ".BN.weight")), raw(st(cla_p, ".BN.bias"))
        cw1f, cb1f = _fold(c_w1, c_b1, c_ln_g, c_ln_b)
        s_l1, s_l1_t = _Stack(self.keep, cw1f, cb1f, P), _Stack(self.keep, _t(cw1f), None, P)
        s_l2, s_l2_t = _Stack(self.keep, c_w2, c_b2, P), _Stack(self.keep, _t(c_w2), None, P)
        s_l3, s_l3_t = _Stack(self.keep, c_w3, c_b3, P), _Stack(self.keep, _t(c_w3 * c_ls[:, :, None]), None, P)
        self.cla = []
        for i, p in enumerate(cla_p):
            tw = L.ClaTW(l1=s_l1.lin(i), l1_t=s_l1_t.lin(i), dw_w=at(c_dwt, i), dw_wf=at(c_dwf, i), dw_b=at(c_db, i), zeros=self.zeros.data_ptr(),
                         l2=s_l2.lin(i), l2_t=s_l2_t.lin(i), bn_g=at(c_bn_g, i), bn_b=at(c_bn_b, i),
                         bn_rm=sd[p + ".BN.running_mean"].data_ptr(), bn_rv=sd[p + ".BN.running_var"].data_ptr(),
                         l3=s_l3.lin(i), l3_t=s_l3_t.lin(i), ls=at(c_ls, i), w1=at(c_w1, i), ln_g=at(c_ln_g, i), ln_b=at(c_ln_b, i),
                         w3=at(c_w3, i), b3=at(c_b3, i), seed_salt=self.salt.data_ptr())
            gr = L.ClaGrad(ln_g=gp(p + ".layer_norm.weight"), ln_b=gp(p + ".layer_norm.bias"), w1=gp(p + ".linear1.weight"),
                           b1=gp(p + ".linear1.bias"), dw_w=gp(p + ".dw_conv_1d.weight"), dw_b=gp(p + ".dw_conv_1d.bias"),
                           w2=gp(p + ".linear2.weight"), b2=gp(p + ".linear2.bias"), bn_g=gp(p + ".BN.weight"), bn_b=gp(p + ".BN.bias"),
                           w3=gp(p + ".linear3.1.weight"), b3=gp(p + ".linear3.1.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
            self.cla.append((tw, gr))

        # ---- DownConv ------------------------------------------------------------------------------------------------
        Kd = cfg.down_kernel
        d_w = raw(st(down_p, ".down_conv.weight", view=(F, Kd)).transpose(1, 2))            # [G,K,F]
        d_b = raw(st(down_p, ".down_conv.bias"))
        d_g, d_bb = raw(st(down_p, ".BN.weight")), raw(st(down_p, ".BN.bias"))
        self.down = []
        for i, p in enumerate(down_p):
            tw = L.DownTW(w=at(d_w, i), b=at(d_b, i), bn_g=at(d_g, i), bn_b=at(d_bb, i), bn_rm=sd[p + ".BN.running_mean"].data_ptr(),
                          bn_rv=sd[p + ".BN.running_var"].data_ptr())
            gr = L.DownGrad(w=gp(p + ".down_conv.weight"), b=gp(p + ".down_conv.bias"), bn_g=gp(p + ".BN.weight"), bn_b=gp(p + ".BN.bias"))
            self.down.append((tw, gr))

        # ---- SpkSplit ------------------------------------------------------------------------------------------------
        sp_w1, sp_b1 = st(split_p, ".linear.0.weight", view=(4 * F * S, F)), st(split_p, ".linear.0.bias")
        sp_w2, sp_b2 = st(split_p, ".linear.2.weight", view=(F * S, 2 * F * S)), st(split_p, ".linear.2.bias")
        sp_g, sp_b = raw(st(split_p, ".norm.weight")), raw(st(split_p, ".norm.bias"))
        s1, s1t = _Stack(self.keep, sp_w1, sp_b1, P), _Stack(self.keep, _t(sp_w1), None, P)
        s2, s2t = _Stack(self.keep, sp_w2, sp_b2, P), _Stack(self.keep, _t(sp_w2), None, P)
        splits = []
        for i, p in enumerate(split_p):
            tw = L.SplitTW(l1=s1.lin(i), l1_t=s1t.lin(i), l2=s2.lin(i), l2_t=s2t.lin(i), gn_g=at(sp_g, i), gn_b=at(sp_b, i))
            gr = L.SplitGrad(w1=gp(p + ".linear.0.weight"), b1=gp(p + ".linear.0.bias"), w2=gp(p + ".linear.2.weight"),
                             b2=gp(p + ".linear.2.bias"), gn_g=gp(p + ".norm.weight"), gn_b=gp(p + ".norm.bias"))
            splits.append((tw, gr))
        self.splits = splits if cfg.per_level_split else splits * (R + 1)

        # ---- fusion ----------------------------------------------------------------------------------------------------
        f_w, f_b = st(fuse_p, ".weight", view=(F, 2 * F)), st(fuse_p, ".bias")
        sf, sft = _Stack(self.keep, f_w, f_b, P), _Stack(self.keep, _t(f_w), None, P)
        self.fuse = [(L.FuseTW(l=sf.lin(i), l_t=sft.lin(i)), L.FuseGrad(w=gp(p + ".weight"), b=gp(p + ".bias"))) for i, p in enumerate(fuse_p)]

        # ---- output heads ------------------------------------------------------------------------------------------------
        o_w1, o_b1 = st(out_p, ".end_conv1x1.0.weight"), st(out_p, ".end_conv1x1.0.bias")
        o_w2, o_b2 = st(out_p, ".end_conv1x1.2.weight"), st(out_p, ".end_conv1x1.2.bias")
        Kc = cfg.enc_kernel
        o_dec = raw(torch.stack([sd[n].reshape(N, Kc) for n in dec_w], 0).to(torch.float32).transpose(1, 2))   # [G,K,N]
        so1, so1t = _Stack(self.keep, o_w1, o_b1, P), _Stack(self.keep, _t(o_w1), None, P)
        so2, so2t = _Stack(self.keep, o_w2, o_b2, P), _Stack(self.keep, _t(o_w2), None, P)
        self.outs = []
        for i, p in enumerate(out_p):
            tw = L.OutTW(l1=so1.lin(i), l1_t=so1t.lin(i), l2=so2.lin(i), l2_t=so2t.lin(i), wdec=at(o_dec, i))
            gr = L.OutGrad(w1=gp(p + ".end_conv1x1.0.weight"), b1=gp(p + ".end_conv1x1.0.bias"), w2=gp(p + ".end_conv1x1.2.weight"),
                           b2=gp(p + ".end_conv1x1.2.bias"), wdec=gp(dec_w[i]))
            self.outs.append((tw, gr))

        # ---- encoder + projector ---------------------------------------------------------------------------------------
        w_enc = raw(sd["audio_encoder.conv1d.weight"].reshape(N, Kc).to(torch.float32).t()[None])                 # [1,K,N]
        pr_w = raw(sd["feature_projector.conv1d.weight"].reshape(1, F, N).to(torch.float32))
        pr_g, pr_b = raw(sd["feature_projector.norm.weight"].to(torch.float32)[None]), raw(sd["feature_projector.norm.bias"].to(torch.float32)[None])
        s_pt = _Stack(self.keep, _t((pr_w.double() * pr_g.double()[:, None, :]).float()), None, P)
        self.front = (L.FrontTW(w_enc=w_enc.data_ptr(), proj_w=pr_w.data_ptr(), gn_g=pr_g.data_ptr(), gn_b=pr_b.data_ptr(), proj_t=s_pt.lin(0),
                                ones=self.ones.data_ptr()),
                      L.FrontGrad(w_enc=gp("audio_encoder.conv1d.weight"), gn_g=gp("feature_projector.norm.weight"),
                                  gn_b=gp("feature_projector.norm.bias"), proj_w=gp("feature_projector.conv1d.weight")))

        # ---- topology views ----------------------------------------------------------------------------------------------
        def resolve(ix):
            return {"g": [(self.ega[a], self.gcfn[b]) for a, b in ix["g"]], "l": [(self.cla[a], self.gcfn[b]) for a, b in ix["l"]],
                    "spk": [(self.spk[a], self.gcfn[b]) for a, b in ix["spk"]]}

        self.enc_stages = [resolve(ix) for ix in enc_idx]
        for i in range(R):
            self.enc_stages[i]["down"] = self.down[i]
        self.bottleneck = resolve(bott_idx)
        self.dec_stages = [resolve(ix) for ix in dec_idx]
        self.out_main, self.out_aux = self.outs[0], self.outs[1:]
        self.bn_counters = [sd[p + ".BN.num_batches_tracked"] for p in cla_p + down_p]
