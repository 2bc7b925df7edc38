"""Per-step weight forms of the training path (``include/sepr.h``, "Training path").

Every optimizer step changes every weight, so these forms are rebuilt once per training forward.  To keep that cheap the
blocks are packed BY TYPE: all 56 GCFN up-projections (22 EGA q/k/v, 22 CLA linear1, ...) are ONE launch of the device-side
re-pack (``sepr_train_pack_lin``: fold, transpose and bf16 split straight from the parameters through a pointer table; round 5 -
rounds 2-4 used ~12 batched torch ops per stack, 870 aten launches and 4 ms per captured step); a block's struct then just
points at its slice.  What is built per projection ``y = norm_affine(x) . W^T + b`` (``sepr_lin``):

* forward form: ``W * gamma`` and ``b + W . beta`` (LayerNorm / GroupNorm affine folded in fp64, like ``pack.py``);
* input-gradient form: the transpose ``(W * gamma)^T``, and for a projection followed by LayerScale ``(ls * W)^T``;
* exact-f32 mode keeps fp32 ``[N,K]`` matrices, bf16x3 mode the ``pack_x3`` fragments (hi/lo bf16 planes); ``bf16`` mode
  (plain bf16 operands, one MFMA per product, fp32 accumulate and fp32 master weights: the training precision BASELINE
  configs[4] names) uses the same fragments with ``sepr_lin.planes = 1`` (the kernels then read the hi plane only);
* GCFN blocks additionally get the fused kernel's weight forms (``pack.pack_gcfn_fused_batched``) when F is 64 or 128: their
  train forward is then one launch that keeps only the LayerNorm statistics (``include/sepr.h`` sepr_gcfn_tw).

The raw parameters ride along for the gradient finishers (read IN PLACE: every parameter is a contiguous fp32 tensor), and a ``*Grad`` struct per block points into one flat fp32
gradient buffer laid out exactly like the parameters (``GradBuffer``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import os
import sys

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


_TABLES: Dict[tuple, torch.Tensor] = {}      # device pointer tables, keyed by (device index, the pointers): built once per weight placement


def _table(tensors: List[torch.Tensor], keep: Optional[list] = None) -> torch.Tensor:
    """Device table of the tensors' addresses (int64).  Cached: the parameters of a model do not move between steps, so the
    per-step re-pack issues no host-to-device copy (a captured step could not contain one); ``model.to()`` changes the key.
    The table's ADDRESS is baked into captured hipGraphs (the pack launches' arguments), so the pack that uses a table holds it
    (``keep``), and the cache only ever evicts tables nobody else references (round 6: a global clear could free a table a live
    ``CapturedTrainStep`` replays against)."""
    dev = tensors[0].device
    key = (dev.index, tuple(t.data_ptr() for t in tensors))
    tab = _TABLES.get(key)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("TrainPack: pointer table missing during a hipGraph capture (run one eager training forward first)")
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or t.data_ptr() % 16:
                raise ValueError("TrainPack reads the parameters in place: every tensor must be contiguous, 16-byte aligned fp32 on one device")
        if len(_TABLES) > 4096:
            for k in [k for k, v in _TABLES.items() if sys.getrefcount(v) <= 3]:      # the dict, the loop variable, getrefcount's argument
                del _TABLES[k]
        tab = _TABLES[key] = torch.tensor(key[1], dtype=torch.int64, device=dev)
    if keep is not None:
        keep.append(tab)
    return tab


class _Stack:
    """A stack of G same-shaped projections in one precision, laid out ON THE DEVICE by one ``sepr_train_pack_lin`` launch (+ one
    ``sepr_train_fold_bias`` launch for the bias) straight from the parameters (``include/sepr.h``, "per-step weight re-pack").

    ``w``: G lists of ``panels`` tensors - block g's ``[SN, SK]`` source matrix is their row-wise concatenation (q / k / v: 3 panels);
    ``b``: the matching bias tensors (or None); ``scale`` / ``scale_kind``: per-block fold factor, 1 = per source column (LayerNorm /
    GroupNorm gamma), 2 = per source row (LayerScale); ``beta``: LayerNorm beta folded into the bias (fp64); ``transpose``: the
    input-gradient form.  ``lin(i)`` is block i's ``sepr_lin``.  Rounds 2-4 built the same bytes with ~12 batched torch ops per stack."""

    def __init__(self, keep: list, w: List[List[torch.Tensor]], shape: Tuple[int, int], precision: str, b: Optional[List[List[torch.Tensor]]] = None,
                 scale: Optional[List[torch.Tensor]] = None, scale_kind: int = 0, beta: Optional[List[torch.Tensor]] = None,
                 transpose: bool = False):
        lib = L.load()
        G, panels = len(w), len(w[0])
        SN, SK = shape
        if any(len(ws) != panels or sum(t.numel() for t in ws) != SN * SK for ws in w):
            raise ValueError("TrainPack: a projection stack needs same-shaped blocks")
        self.G, self.N, self.K = G, (SK if transpose else SN), (SN if transpose else SK)
        if self.N % 16 or self.K % 32:
            raise ValueError(f"the packed projection forms need N % 16 == 0 and K % 32 == 0, got {self.N}x{self.K}")
        dev = w[0][0].device
        st = torch.cuda.current_stream(dev).cuda_stream
        flat_w = [t for ws in w for t in ws]
        wt = _table(flat_w, keep)
        self.planes = 1 if precision == "bf16" else 0
        packed = precision in ("bf16x3", "bf16")
        if packed:
            self.wp = torch.empty(G, 2 * self.N * self.K, dtype=torch.bfloat16, device=dev)
            self.w = None
            out = self.wp
        else:
            self.w = torch.empty(G, self.N, self.K, dtype=torch.float32, device=dev)
            self.wp = None
            out = self.w
        keep.append(out)
        sct = _table(scale, keep).data_ptr() if scale_kind else None
        L.check(lib.sepr_train_pack_lin(wt.data_ptr(), sct, G, SN, SK, panels, scale_kind, 1 if transpose else 0, 1 if packed else 0,
                                        out.data_ptr(), st), "sepr_train_pack_lin")
        self.b = None
        if b is not None or beta is not None:
            if transpose:
                raise ValueError("a transposed (input-gradient) form carries no bias")
            self.b = torch.empty(G, self.N, dtype=torch.float32, device=dev)
            keep.append(self.b)
            bt = _table([t for bs in b for t in bs], keep).data_ptr() if b is not None else None
            L.check(lib.sepr_train_fold_bias(wt.data_ptr() if beta is not None else None, bt, _table(beta, keep).data_ptr() if beta is not None else None,
                                             G, self.N, self.K, panels, self.b.data_ptr(), st), "sepr_train_fold_bias")

    def lin(self, i: int) -> L.Lin:
        bptr = None if self.b is None else self.b.data_ptr() + 4 * i * self.N
        if self.wp is not None:
            return L.Lin(w=None, wp=self.wp.data_ptr() + 2 * i * 2 * self.N * self.K, b=bptr, planes=self.planes)
        return L.Lin(w=self.w.data_ptr() + 4 * i * self.N * self.K, wp=None, b=bptr, planes=0)


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

        def raw(t):       # keep a contiguous fp32 stack alive
            t = t.contiguous()
            self.keep.append(t)
            return t

        def at(t, i):
            return t.data_ptr() + 4 * i * (t.numel() // t.shape[0])

        def ps(names, suffix):          # the parameters themselves (read in place by the device-side re-pack and by the gradient finishers)
            return [sd[n + suffix] for n in names]

        def ptr(name):
            return sd[name].data_ptr()

        def one(ts):                    # G single-panel blocks
            return [[t] for t in ts]

        gp, P, keep = grads.ptr, precision, self.keep
        # ---- GCFN --------------------------------------------------------------------------------------------------
        g_w1, g_b1 = ps(gcfn_p, ".net1.1.weight"), ps(gcfn_p, ".net1.1.bias")
        g_w2, g_b2 = ps(gcfn_p, ".net2.2.weight"), ps(gcfn_p, ".net2.2.bias")
        g_lg, g_lb = ps(gcfn_p, ".net1.0.weight"), ps(gcfn_p, ".net1.0.bias")
        g_ls = ps(gcfn_p, ".Layer_scale.layer_scale")
        g_dw = raw(st(gcfn_p, ".depthwise.weight", view=(6 * F, 3)).transpose(1, 2))          # [G,3,6F] tap-major
        s_up = _Stack(keep, one(g_w1), (6 * F, F), P, b=one(g_b1), scale=g_lg, scale_kind=1, beta=g_lb)
        s_up_t = _Stack(keep, one(g_w1), (6 * F, F), P, scale=g_lg, scale_kind=1, transpose=True)
        s_dn = _Stack(keep, one(g_w2), (F, 3 * F), P, b=one(g_b2))
        s_dn_t = _Stack(keep, one(g_w2), (F, 3 * F), P, scale=g_ls, scale_kind=2, transpose=True)
        # fused forward (+ statistics-only context) and recomputing backward: packed-bf16 precisions, F in {64, 128}
        self.fused_gcfn = (P in ("bf16x3", "bf16") and F in (64, 128) and os.environ.get("SEPR_TRAIN_FUSE_GCFN", "1") != "0")
        fw1 = fw2 = None
        if self.fused_gcfn:
            # the fused kernel's forms of all blocks in ONE launch, from the parameters in place (sepr_train_pack_gcfn_fused; the batched torch
            # formulation pack.pack_gcfn_fused_batched remains the inference packer's and the device test's reference)
            G_, KS_, nch_ = len(gcfn_p), F // 32, 3 * F // 32
            fw1 = torch.empty(G_, nch_ * (4 * KS_ * 2048 + 4096), dtype=torch.uint8, device=dev)
            fw2 = torch.empty(G_, nch_, F // 16, 2, 64, 8, dtype=torch.bfloat16, device=dev)
            tb = lambda ts: _table(ts, self.keep).data_ptr()                                    # noqa: E731
            L.check(L.load().sepr_train_pack_gcfn_fused(tb(g_w1), tb(g_b1), tb(g_lg), tb(g_lb), tb(g_w2), tb(ps(gcfn_p, ".depthwise.weight")),
                                                        tb(ps(gcfn_p, ".depthwise.bias")), G_, F, fw1.data_ptr(), fw2.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream), "sepr_train_pack_gcfn_fused")
            self.keep += [fw1, fw2]
        self.gcfn = []
        for i, p in enumerate(gcfn_p):
            tw = L.GcfnTW(up=s_up.lin(i), up_t=s_up_t.lin(i), down=s_dn.lin(i), down_t=s_dn_t.lin(i), dw_w=at(g_dw, i), dw_b=ptr(p + ".depthwise.bias"),
                          ls=ptr(p + ".Layer_scale.layer_scale"), w1=ptr(p + ".net1.1.weight"), ln_g=ptr(p + ".net1.0.weight"),
                          ln_b=ptr(p + ".net1.0.bias"), w2=ptr(p + ".net2.2.weight"), b2=ptr(p + ".net2.2.bias"),
                          fused_w1p=None if fw1 is None else fw1.data_ptr() + i * fw1.shape[1],
                          fused_w2p=None if fw2 is None else fw2.data_ptr() + i * fw2[0].numel() * 2, seed_salt=self.salt.data_ptr())
            gr = L.GcfnGrad(ln_g=gp(p + ".net1.0.weight"), ln_b=gp(p + ".net1.0.bias"), w1=gp(p + ".net1.1.weight"), b1=gp(p + ".net1.1.bias"),
                            dw_w=gp(p + ".depthwise.weight"), dw_b=gp(p + ".depthwise.bias"), w2=gp(p + ".net2.2.weight"),
                            b2=gp(p + ".net2.2.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
            self.gcfn.append((tw, gr))

        # ---- MHA (EGA's and SpkAttention's) ------------------------------------------------------------------------
        def mha_stack(names):
            qkv_w = [[sd[n + f".linear_{c}.weight"] for c in "qkv"] for n in names]
            qkv_b = [[sd[n + f".linear_{c}.bias"] for c in "qkv"] for n in names]
            ln_g, ln_b = ps(names, ".layer_norm.weight"), ps(names, ".layer_norm.bias")
            wo, bo, ls = ps(names, ".linear_out.weight"), ps(names, ".linear_out.bias"), ps(names, ".Layer_scale.layer_scale")
            s_qkv = _Stack(keep, qkv_w, (3 * F, F), P, b=qkv_b, scale=ln_g, scale_kind=1, beta=ln_b)
            s_qkv_t = _Stack(keep, qkv_w, (3 * F, F), P, scale=ln_g, scale_kind=1, transpose=True)
            s_out = _Stack(keep, one(wo), (F, F), P, b=one(bo))
            s_out_t = _Stack(keep, one(wo), (F, F), P, scale=ls, scale_kind=2, transpose=True)
            wqkv = _Stack(keep, qkv_w, (3 * F, F), "fp32")                                    # raw stacked [3F,F] for the gradient finisher
            res = []
            for i, p in enumerate(names):
                tw = L.MhaTW(qkv=s_qkv.lin(i), qkv_t=s_qkv_t.lin(i), out=s_out.lin(i), out_t=s_out_t.lin(i), ls=ptr(p + ".Layer_scale.layer_scale"),
                             wqkv=wqkv.lin(i).w, ln_g=ptr(p + ".layer_norm.weight"), ln_b=ptr(p + ".layer_norm.bias"), wo=ptr(p + ".linear_out.weight"),
                             bo=ptr(p + ".linear_out.bias"), seed_salt=self.salt.data_ptr())
                gr = L.MhaGrad(ln_g=gp(p + ".layer_norm.weight"), ln_b=gp(p + ".layer_norm.bias"),
                               wq=gp(p + ".linear_q.weight"), bq=gp(p + ".linear_q.bias"), wk=gp(p + ".linear_k.weight"),
                               bk=gp(p + ".linear_k.bias"), wv=gp(p + ".linear_v.weight"), bv=gp(p + ".linear_v.bias"),
                               wo=gp(p + ".linear_out.weight"), bo=gp(p + ".linear_out.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
                res.append((tw, gr))
            return res

        ega_mha = mha_stack([p + ".block.self_attn" for p in ega_p])
        self.spk = mha_stack(spk_p) if spk_p else []
        # ---- EGA gate ------------------------------------------------------------------------------------------------
        e_w, e_b = ps(ega_p, ".block.linear.1.weight"), ps(ega_p, ".block.linear.1.bias")
        e_lg, e_lb = ps(ega_p, ".block.linear.0.weight"), ps(ega_p, ".block.linear.0.bias")
        s_gate = _Stack(keep, one(e_w), (F, F), P, b=one(e_b), scale=e_lg, scale_kind=1, beta=e_lb)
        s_gate_t = _Stack(keep, one(e_w), (F, F), P, scale=e_lg, scale_kind=1, transpose=True)
        pe = sd["separator.pos_emb.pe_k.weight"]
        self.ega = []
        for i, p in enumerate(ega_p):
            tw = L.EgaTW(attn=ega_mha[i][0], gate=s_gate.lin(i), gate_t=s_gate_t.lin(i), gate_w=ptr(p + ".block.linear.1.weight"),
                         gate_ln_g=ptr(p + ".block.linear.0.weight"), gate_ln_b=ptr(p + ".block.linear.0.bias"), pe_k=pe.data_ptr(), maxlen=cfg.maxlen)
            gr = L.EgaGrad(attn=ega_mha[i][1], gate_ln_g=gp(p + ".block.linear.0.weight"), gate_ln_b=gp(p + ".block.linear.0.bias"),
                           gate_w=gp(p + ".block.linear.1.weight"), gate_b=gp(p + ".block.linear.1.bias"),
                           pe_k=gp("separator.pos_emb.pe_k.weight"))
            self.ega.append((tw, gr))

        # ---- CLA -----------------------------------------------------------------------------------------------------
        K = cfg.cla_kernel
        c_w1, c_b1 = ps(cla_p, ".linear1.weight"), ps(cla_p, ".linear1.bias")
        c_w2, c_b2 = ps(cla_p, ".linear2.weight"), ps(cla_p, ".linear2.bias")
        c_w3, c_b3 = ps(cla_p, ".linear3.1.weight"), ps(cla_p, ".linear3.1.bias")
        c_lg, c_lb, c_ls = ps(cla_p, ".layer_norm.weight"), ps(cla_p, ".layer_norm.bias"), ps(cla_p, ".Layer_scale.layer_scale")
        c_dw = st(cla_p, ".dw_conv_1d.weight", view=(F, K))
        c_dwt, c_dwf = raw(c_dw.transpose(1, 2)), raw(c_dw.flip(2).transpose(1, 2))          # [G,K,F] tap-major / reversed taps
        s_l1 = _Stack(keep, one(c_w1), (2 * F, F), P, b=one(c_b1), scale=c_lg, scale_kind=1, beta=c_lb)
        s_l1_t = _Stack(keep, one(c_w1), (2 * F, F), P, scale=c_lg, scale_kind=1, transpose=True)
        s_l2, s_l2_t = _Stack(keep, one(c_w2), (2 * F, F), P, b=one(c_b2)), _Stack(keep, one(c_w2), (2 * F, F), P, transpose=True)
        s_l3 = _Stack(keep, one(c_w3), (F, 2 * F), P, b=one(c_b3))
        s_l3_t = _Stack(keep, one(c_w3), (F, 2 * F), P, scale=c_ls, scale_kind=2, transpose=True)
        self.cla = []
        for i, p in enumerate(cla_p):
            tw = L.ClaTW(l1=s_l1.lin(i), l1_t=s_l1_t.lin(i), dw_w=at(c_dwt, i), dw_wf=at(c_dwf, i), dw_b=ptr(p + ".dw_conv_1d.bias"), zeros=self.zeros.data_ptr(),
                         l2=s_l2.lin(i), l2_t=s_l2_t.lin(i), bn_g=ptr(p + ".BN.weight"), bn_b=ptr(p + ".BN.bias"),
                         bn_rm=sd[p + ".BN.running_mean"].data_ptr(), bn_rv=sd[p + ".BN.running_var"].data_ptr(),
                         l3=s_l3.lin(i), l3_t=s_l3_t.lin(i), ls=ptr(p + ".Layer_scale.layer_scale"), w1=ptr(p + ".linear1.weight"),
                         ln_g=ptr(p + ".layer_norm.weight"), ln_b=ptr(p + ".layer_norm.bias"), w3=ptr(p + ".linear3.1.weight"), b3=ptr(p + ".linear3.1.bias"),
                         seed_salt=self.salt.data_ptr())
            gr = L.ClaGrad(ln_g=gp(p + ".layer_norm.weight"), ln_b=gp(p + ".layer_norm.bias"), w1=gp(p + ".linear1.weight"),
                           b1=gp(p + ".linear1.bias"), dw_w=gp(p + ".dw_conv_1d.weight"), dw_b=gp(p + ".dw_conv_1d.bias"),
                           w2=gp(p + ".linear2.weight"), b2=gp(p + ".linear2.bias"), bn_g=gp(p + ".BN.weight"), bn_b=gp(p + ".BN.bias"),
                           w3=gp(p + ".linear3.1.weight"), b3=gp(p + ".linear3.1.bias"), ls=gp(p + ".Layer_scale.layer_scale"))
            self.cla.append((tw, gr))

        # ---- DownConv ------------------------------------------------------------------------------------------------
        Kd = cfg.down_kernel
        d_w = raw(st(down_p, ".down_conv.weight", view=(F, Kd)).transpose(1, 2))            # [G,K,F]
        self.down = []
        for i, p in enumerate(down_p):
            tw = L.DownTW(w=at(d_w, i), b=ptr(p + ".down_conv.bias"), bn_g=ptr(p + ".BN.weight"), bn_b=ptr(p + ".BN.bias"),
                          bn_rm=sd[p + ".BN.running_mean"].data_ptr(), bn_rv=sd[p + ".BN.running_var"].data_ptr())
            gr = L.DownGrad(w=gp(p + ".down_conv.weight"), b=gp(p + ".down_conv.bias"), bn_g=gp(p + ".BN.weight"), bn_b=gp(p + ".BN.bias"))
            self.down.append((tw, gr))

        # ---- SpkSplit ------------------------------------------------------------------------------------------------
        sp_w1, sp_b1 = ps(split_p, ".linear.0.weight"), ps(split_p, ".linear.0.bias")        # conv1x1 weights [N,K,1]: the [N,K] matrix in place
        sp_w2, sp_b2 = ps(split_p, ".linear.2.weight"), ps(split_p, ".linear.2.bias")
        s1, s1t = _Stack(keep, one(sp_w1), (4 * F * S, F), P, b=one(sp_b1)), _Stack(keep, one(sp_w1), (4 * F * S, F), P, transpose=True)
        s2, s2t = _Stack(keep, one(sp_w2), (F * S, 2 * F * S), P, b=one(sp_b2)), _Stack(keep, one(sp_w2), (F * S, 2 * F * S), P, transpose=True)
        splits = []
        for i, p in enumerate(split_p):
            tw = L.SplitTW(l1=s1.lin(i), l1_t=s1t.lin(i), l2=s2.lin(i), l2_t=s2t.lin(i), gn_g=ptr(p + ".norm.weight"), gn_b=ptr(p + ".norm.bias"))
            gr = L.SplitGrad(w1=gp(p + ".linear.0.weight"), b1=gp(p + ".linear.0.bias"), w2=gp(p + ".linear.2.weight"),
                             b2=gp(p + ".linear.2.bias"), gn_g=gp(p + ".norm.weight"), gn_b=gp(p + ".norm.bias"))
            splits.append((tw, gr))
        self.splits = splits if cfg.per_level_split else splits * (R + 1)

        # ---- fusion ----------------------------------------------------------------------------------------------------
        f_w, f_b = ps(fuse_p, ".weight"), ps(fuse_p, ".bias")
        sf, sft = _Stack(keep, one(f_w), (F, 2 * F), P, b=one(f_b)), _Stack(keep, one(f_w), (F, 2 * F), P, transpose=True)
        self.fuse = [(L.FuseTW(l=sf.lin(i), l_t=sft.lin(i)), L.FuseGrad(w=gp(p + ".weight"), b=gp(p + ".bias"))) for i, p in enumerate(fuse_p)]

        # ---- output heads ------------------------------------------------------------------------------------------------
        o_w1, o_b1 = ps(out_p, ".end_conv1x1.0.weight"), ps(out_p, ".end_conv1x1.0.bias")
        o_w2, o_b2 = ps(out_p, ".end_conv1x1.2.weight"), ps(out_p, ".end_conv1x1.2.bias")
        Kc = cfg.enc_kernel
        o_dec = raw(torch.stack([sd[n].reshape(N, Kc) for n in dec_w], 0).to(torch.float32).transpose(1, 2))   # [G,K,N]
        so1, so1t = _Stack(keep, one(o_w1), (4 * F, F), P, b=one(o_b1)), _Stack(keep, one(o_w1), (4 * F, F), P, transpose=True)
        so2, so2t = _Stack(keep, one(o_w2), (N, 2 * F), P, b=one(o_b2)), _Stack(keep, one(o_w2), (N, 2 * F), P, transpose=True)
        self.outs = []
        for i, p in enumerate(out_p):
            tw = L.OutTW(l1=so1.lin(i), l1_t=so1t.lin(i), l2=so2.lin(i), l2_t=so2t.lin(i), wdec=at(o_dec, i))
            gr = L.OutGrad(w1=gp(p + ".end_conv1x1.0.weight"), b1=gp(p + ".end_conv1x1.0.bias"), w2=gp(p + ".end_conv1x1.2.weight"),
                           b2=gp(p + ".end_conv1x1.2.bias"), wdec=gp(dec_w[i]))
            self.outs.append((tw, gr))

        # ---- encoder + projector ---------------------------------------------------------------------------------------
        w_enc = raw(sd["audio_encoder.conv1d.weight"].reshape(N, Kc).to(torch.float32).t()[None])                 # [1,K,N]
        pr_w, pr_g = sd["feature_projector.conv1d.weight"], sd["feature_projector.norm.weight"]
        s_pt = _Stack(keep, [[pr_w]], (F, N), P, scale=[pr_g], scale_kind=1, transpose=True)           # (W * gamma)^T [N,F]
        self.front = (L.FrontTW(w_enc=w_enc.data_ptr(), proj_w=pr_w.data_ptr(), gn_g=pr_g.data_ptr(), gn_b=ptr("feature_projector.norm.bias"),
                                proj_t=s_pt.lin(0), ones=self.ones.data_ptr()),
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
