"""Training driver: train-mode forward that records a tape, and the backward that replays it in reverse.

Topology restated from reference ``modules/module.py:190-218`` and ``model.py:38-54`` exactly as in ``engine.py`` (the
inference driver); what differs is that every block goes through its ``sepr_*_train_fwd`` entry point, which keeps what
the block's ``sepr_*_bwd`` needs in a context buffer, and that BatchNorm uses batch statistics / dropout is live
(reference ``engine.py:50-83`` runs ``model.train()``).  PyTorch supplies the caching allocator, the current stream and the
``autograd.Function`` shell around this (``model.py``); the gradient flow between blocks is explicit here - no autograd
graph is built per block.

Gradients of all parameters land in one flat fp32 ``GradBuffer`` (zeroed per step); tensors that are read by two consumers
(stage outputs feeding both the skip split and the DownConv; decoder stage inputs feeding both the fusion conv and an
auxiliary head) get their second contribution through the ``*_accumulate`` switches of the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .config import SepConfig
from .engine import nearest_index
from .pack import GN_EPS
from .train_pack import GradBuffer, TrainPack


def inverse_index_start(idx: np.ndarray, src: int) -> np.ndarray:
    """``start[s]`` = first output frame whose nearest source frame is ``s`` (``idx`` is non-decreasing); ``start[src]`` = len."""
    return np.searchsorted(idx, np.arange(src + 1), side="left").astype(np.int32)


class TrainEngine:
    def __init__(self, cfg: SepConfig, device: torch.device):
        self.cfg, self.device = cfg, device
        self.lib = L.load()
        self._ws: Optional[torch.Tensor] = None
        self._idx: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._sizes: Dict[tuple, int] = {}      # sepr_train_ctx_bytes / sepr_train_ws_bytes per (kind, op, shape): one C call each, ever
        self._fin_arena: Optional[torch.Tensor] = None      # deferred-finisher arena (backward)
        self._defer = os.environ.get("SEPR_TRAIN_DEFER", "1") != "0"           # A/B switch of the deferred finishers
        # riding reductions (include/sepr.h sepr_train_defer_parts; needs the deferred finishers): built, bit-identical, measured NOT faster - the
        # appended blocks occupy full workgroup slots of the contraction kernel and a reduction is 33 MB of traffic wherever it runs
        # (profiles/r06_wgrad_dma.txt) - off by default, SEPR_TRAIN_RIDE=1 enables
        self._ride = os.environ.get("SEPR_TRAIN_RIDE", "0") == "1"
        self._parts: Optional[torch.Tensor] = None          # double buffer of partial tiles (riding reductions)
        # Weight-gradient side stream (round 6, include/sepr.h sepr_train_wgrad_stream; SEPR_TRAIN_WGRAD_STREAM=1, OFF by default): inside backward()
        # the contractions run on a second stream beside the input-gradient chain; two workspaces alternate between consecutive block calls (a
        # call's workspace is still read by its side-stream contractions after it returns).  Bit-identical gradients, all 149 training tests
        # green with it on - and MEASURED NOT FASTER: 211.9 / 213.3 vs 214.4 / 213.8 utt/s captured, 195 vs 211 eager (profiles/r06_train_wgrad_stream.txt):
        # every contraction and every kernel of the input-gradient chain fills the machine on its own, two streams only interleave them.
        self._wg_on = os.environ.get("SEPR_TRAIN_WGRAD_STREAM", "0") == "1"
        self._wg_side: Optional[torch.cuda.Stream] = None
        self._wg_active = False
        self._ws_alt: list = [None, None]
        self._ws_slot = 0
        self._wg_keep: list = []                 # tensors a side-stream kernel may still read (the last few d(activation) tensors)
        self._attn_valu = os.environ.get("SEPR_TRAIN_ATTN_VALU", "0") == "1"     # (the library latches it at its first EGA call as well)

    # ---- plumbing -------------------------------------------------------------------------------------------------------
    def _workspace(self, nbytes: int):
        if self._wg_active:
            # one call = one block's backward: close the bracket of the workspace the previous call used, take the OTHER one and wait for
            # the side-stream work of the call that last used it (two calls ago)
            lib = self.lib
            st = torch.cuda.current_stream(self.device).cuda_stream
            L.check(lib.sepr_train_wgrad_mark(self._ws_slot), "sepr_train_wgrad_mark")
            self._ws_slot ^= 1
            L.check(lib.sepr_train_wgrad_wait(self._ws_slot, st), "sepr_train_wgrad_wait")
            buf = self._ws_alt[self._ws_slot]
            if buf is None or buf.numel() < nbytes:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("TrainEngine: a backward workspace would have to grow during a hipGraph capture (run one eager step of this shape first)")
                torch.cuda.synchronize(self.device)          # (warm-up only) the side stream may still read the buffers being replaced
                self._ws_alt = [None, None]                   # BOTH grow to the largest request seen: which call lands on which buffer is a matter of parity
                self._ws_alt = [torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device) for _ in range(2)]
                buf = self._ws_alt[self._ws_slot]
            return buf.data_ptr(), buf.numel()
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        return self._ws.data_ptr(), self._ws.numel()

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _ctx(self, op, n, T, Tp=0, H=0):
        c = self.cfg
        key = ("c", op, n, T, Tp)
        nb = self._sizes.get(key)
        if nb is None:
            nb = self._sizes[key] = int(self.lib.sepr_train_ctx_bytes(op, n, T, Tp, c.feat, c.enc_channels, c.num_spks, H or c.heads))
        return torch.empty(nb, dtype=torch.uint8, device=self.device)

    def _wsfor(self, op, n, T, Tp=0, K=0):
        c = self.cfg
        key = ("w", op, n, T, Tp, K)
        nb = self._sizes.get(key)
        if nb is None:
            nb = self._sizes[key] = int(self.lib.sepr_train_ws_bytes(op, n, T, Tp, c.feat, c.enc_channels, c.num_spks, c.heads, K))
        return self._workspace(nb)

    def _index(self, src: int, dst: int):
        key = (src, dst)
        if key not in self._idx:
            idx = nearest_index(src, dst)
            self._idx[key] = (torch.from_numpy(idx).to(self.device), torch.from_numpy(inverse_index_start(idx, src)).to(self.device))
        return self._idx[key]

    def _gcfn_op(self, tw) -> int:
        """Sizing op of a GCFN block; mirrors ``gcfn_is_fused`` / ``gcfn_pl16`` of csrc/sepr_train_api.hip."""
        if not (tw.fused_w1p and self.cfg.feat in (64, 128)):
            return L.TOP_GCFN
        planes16 = tw.up.planes == 1 and self.lib.sepr_knob(L.KNOB_TRAIN_GCFN_PLANES) != 0     # the library's own (latched) reading
        return L.TOP_GCFN_FUSED16 if planes16 else L.TOP_GCFN_FUSED

    def _ega_op(self, tw) -> int:
        """Sizing op of an EGA block: the packed-bf16 precisions run the attention flash-style on the bf16 MFMA (context = one
        log-sum-exp per query row instead of the [Tp, Tp] probabilities); mirrors ``ega_mfma`` of csrc/sepr_train_api.hip."""
        dk = self.cfg.feat // self.cfg.heads
        mfma = bool(tw.attn.qkv.wp) and dk in (16, 32) and not self._attn_valu
        return L.TOP_EGA_X3 if mfma else L.TOP_EGA

    # ---- single blocks (also what the unit tests drive) --------------------------------------------------------------------
    def block_fwd(self, kind: str, xin: torch.Tensor, w, n: int, Tc: int, Tp: int = 0, p_drop: float = 0.0, seed: int = 0):
        """One residual block (``gcfn`` / ``cla`` / ``ega`` / ``spk``) in train mode -> (output, tape record)."""
        c, lib = self.cfg, self.lib
        F, H, S = c.feat, c.heads, c.num_spks
        st = torch.cuda.current_stream(self.device).cuda_stream
        y = torch.empty_like(xin)
        if kind == "gcfn":
            op = self._gcfn_op(w[0])                     # statistics-only context when fused (+ the bf16 rows in the plain-bf16 precision)
            cx = self._ctx(op, n, Tc)
            L.check(lib.sepr_gcfn_train_fwd(xin.data_ptr(), y.data_ptr(), n, Tc, F, C.byref(w[0]), cx.data_ptr(), cx.numel(),
                                            *self._wsfor(op, n, Tc), p_drop, seed, st), "sepr_gcfn_train_fwd")
        elif kind == "cla":
            cx = self._ctx(L.TOP_CLA, n, Tc)
            L.check(lib.sepr_cla_train_fwd(xin.data_ptr(), y.data_ptr(), n, Tc, F, c.cla_kernel, C.byref(w[0]), cx.data_ptr(), cx.numel(),
                                           *self._wsfor(L.TOP_CLA, n, Tc, 0, c.cla_kernel), p_drop, seed, st), "sepr_cla_train_fwd")
        elif kind == "ega":
            op = self._ega_op(w[0])
            cx = self._ctx(op, n, Tc, Tp)
            L.check(lib.sepr_ega_train_fwd(xin.data_ptr(), y.data_ptr(), n, Tc, Tp, F, H, C.byref(w[0]), cx.data_ptr(), cx.numel(),
                                           *self._wsfor(op, n, Tc, Tp), p_drop, seed, st), "sepr_ega_train_fwd")
        elif kind == "spk":
            cx = self._ctx(L.TOP_SPKATTN, n, Tc)
            L.check(lib.sepr_spkattn_train_fwd(xin.data_ptr(), y.data_ptr(), n, S, Tc, F, H, C.byref(w[0]), cx.data_ptr(), cx.numel(),
                                               *self._wsfor(L.TOP_SPKATTN, n, Tc), p_drop, seed, st), "sepr_spkattn_train_fwd")
        else:
            raise ValueError(kind)
        return y, (kind, xin, cx, w, n, Tc, seed, Tp, p_drop)

    def block_bwd(self, rec, dy: torch.Tensor) -> torch.Tensor:
        """Backward of a ``block_fwd`` record: d(output) -> d(input); parameter gradients accumulate into the GradBuffer."""
        c, lib = self.cfg, self.lib
        F, H, S = c.feat, c.heads, c.num_spks
        kind, xin, cx, w, n, Tc, seed, Tp, p_drop = rec
        st = torch.cuda.current_stream(self.device).cuda_stream
        dx = torch.empty_like(xin)
        if kind == "gcfn":
            op = self._gcfn_op(w[0])
            L.check(lib.sepr_gcfn_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, Tc, F, C.byref(w[0]), C.byref(w[1]), cx.data_ptr(),
                                      cx.numel(), *self._wsfor(op, n, Tc), p_drop, seed, st), "sepr_gcfn_bwd")
        elif kind == "cla":
            L.check(lib.sepr_cla_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, Tc, F, c.cla_kernel, C.byref(w[0]), C.byref(w[1]),
                                     cx.data_ptr(), cx.numel(), *self._wsfor(L.TOP_CLA, n, Tc, 0, c.cla_kernel), p_drop, seed, st),
                    "sepr_cla_bwd")
        elif kind == "ega":
            L.check(lib.sepr_ega_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, Tc, Tp, F, H, C.byref(w[0]), C.byref(w[1]),
                                     cx.data_ptr(), cx.numel(), *self._wsfor(self._ega_op(w[0]), n, Tc, Tp), p_drop, seed, st), "sepr_ega_bwd")
        else:
            L.check(lib.sepr_spkattn_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, S, Tc, F, H, C.byref(w[0]), C.byref(w[1]),
                                         cx.data_ptr(), cx.numel(), *self._wsfor(L.TOP_SPKATTN, n, Tc), p_drop, seed, st), "sepr_spkattn_bwd")
        return dx

    def split_fwd(self, xin, w, B, Tc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        y, cx = self._new(B * c.num_spks, Tc, c.feat), self._ctx(L.TOP_SPLIT, B, Tc)
        L.check(lib.sepr_spksplit_train_fwd(xin.data_ptr(), y.data_ptr(), B, c.num_spks, Tc, c.feat, GN_EPS, C.byref(w[0]), cx.data_ptr(),
                                            cx.numel(), *self._wsfor(L.TOP_SPLIT, B, Tc), st), "sepr_spksplit_train_fwd")
        return y, cx

    def split_bwd(self, xin, cx, w, dy, dx, accumulate, B, Tc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(lib.sepr_spksplit_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), 1 if accumulate else 0, B, c.num_spks, Tc, c.feat,
                                      C.byref(w[0]), C.byref(w[1]), cx.data_ptr(), cx.numel(), *self._wsfor(L.TOP_SPLIT, B, Tc), st),
                "sepr_spksplit_bwd")

    def down_fwd(self, xin, w, B, Tc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        K = c.down_kernel
        To = (Tc + 2 * ((K - 1) // 2) - K) // 2 + 1
        y, cx = self._new(B, To, c.feat), self._ctx(L.TOP_DOWN, B, Tc)
        L.check(lib.sepr_downconv_train_fwd(xin.data_ptr(), y.data_ptr(), B, Tc, c.feat, K, C.byref(w[0]), cx.data_ptr(), cx.numel(),
                                            *self._wsfor(L.TOP_DOWN, B, Tc, 0, K), st), "sepr_downconv_train_fwd")
        return y, cx, To

    def down_bwd(self, xin, cx, w, dy, B, Tc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        dx = torch.empty_like(xin)
        L.check(lib.sepr_downconv_bwd(xin.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, Tc, c.feat, c.down_kernel, C.byref(w[0]), C.byref(w[1]),
                                      cx.data_ptr(), cx.numel(), *self._wsfor(L.TOP_DOWN, B, Tc, 0, c.down_kernel), st), "sepr_downconv_bwd")
        return dx

    def fuse_fwd(self, lo, skip, w, nS, Ts):
        st = torch.cuda.current_stream(self.device).cuda_stream
        y = torch.empty_like(skip)
        L.check(self.lib.sepr_fuse_fwd(lo.data_ptr(), skip.data_ptr(), y.data_ptr(), nS, Ts, self.cfg.feat, C.byref(_fuse_fwd_w(w[0])), st),
                "sepr_fuse_fwd")
        return y

    def fuse_bwd(self, lo, skip, w, dy, nS, Ts):
        st = torch.cuda.current_stream(self.device).cuda_stream
        dlo, dsk = torch.empty_like(lo), torch.empty_like(skip)
        L.check(self.lib.sepr_fuse_bwd(lo.data_ptr(), skip.data_ptr(), dy.data_ptr(), dlo.data_ptr(), dsk.data_ptr(), nS, Ts, self.cfg.feat,
                                       C.byref(w[0]), C.byref(w[1]), *self._wsfor(L.TOP_FUSE, nS, Ts), st), "sepr_fuse_bwd")
        return dlo, dsk

    def head_fwd(self, xin, w, nS, Tsrc, L_, idx, enc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        Tout = (L_ - 1) * c.enc_stride + c.enc_kernel
        wav = self._new(c.num_spks, nS // c.num_spks, Tout)
        cx = self._ctx(L.TOP_OUT, nS, L_, Tsrc)
        L.check(lib.sepr_outlayer_decoder_train_fwd(
            xin.data_ptr(), nS, c.num_spks, Tsrc, L_, None if idx is None else idx[0].data_ptr(), None if enc is None else enc.data_ptr(),
            c.feat, c.enc_channels, c.enc_kernel, c.enc_stride, C.byref(w[0]), wav.data_ptr(), cx.data_ptr(), cx.numel(), None, 0, st),
            "sepr_outlayer_decoder_train_fwd")
        return wav, cx

    def head_bwd(self, xin, cx, w, d_wav, dx, accumulate, denc, nS, Tsrc, L_, idx, enc):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(lib.sepr_outlayer_decoder_bwd(
            xin.data_ptr(), d_wav.data_ptr(), dx.data_ptr(), 1 if accumulate else 0, None if denc is None else denc.data_ptr(), nS, c.num_spks,
            Tsrc, L_, None if idx is None else idx[0].data_ptr(), None if idx is None else idx[1].data_ptr(),
            None if enc is None else enc.data_ptr(), c.feat, c.enc_channels, c.enc_kernel, c.enc_stride, C.byref(w[0]), C.byref(w[1]),
            cx.data_ptr(), cx.numel(), *self._wsfor(L.TOP_OUT, nS, L_, Tsrc), st), "sepr_outlayer_decoder_bwd")

    def front_fwd(self, x, tp, B, T, L_, Lp):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        enc, cur = self._new(B, L_, c.enc_channels), self._new(B, Lp, c.feat)
        ctx = self._ctx(L.TOP_FRONT, B, T, Lp)
        L.check(lib.sepr_front_train_fwd(x.data_ptr(), B, T, c.enc_channels, c.enc_kernel, c.enc_stride, c.feat, Lp, GN_EPS, C.byref(tp.front[0]),
                                         enc.data_ptr(), cur.data_ptr(), ctx.data_ptr(), ctx.numel(),
                                         *self._wsfor(L.TOP_FRONT, B, T, Lp), st), "sepr_front_train_fwd")
        return enc, cur, ctx

    def front_bwd(self, x, enc, ctx, tp, dcur, denc, B, T, Lp):
        c, lib = self.cfg, self.lib
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(lib.sepr_front_bwd(x.data_ptr(), enc.data_ptr(), dcur.data_ptr(), None if denc is None else denc.data_ptr(), B, T, c.enc_channels,
                                   c.enc_kernel, c.enc_stride, c.feat, Lp, C.byref(tp.front[0]), C.byref(tp.front[1]), ctx.data_ptr(),
                                   ctx.numel(), *self._wsfor(L.TOP_FRONT, B, T, Lp), st), "sepr_front_bwd")

    # ---- forward ----------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, tp: TrainPack, p_drop: float, seed: int, with_aux: bool = True):
        """x ``[B,T]`` -> (wav ``[S,B,T']``, list of R aux ``[S,B,T']``, tape)."""
        c, lib = self.cfg, self.lib
        if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
            raise RuntimeError("separator input must be a float32 [batch, samples] tensor on the HIP device")
        x = x.contiguous()
        B, T = x.shape
        S, R, F, N, H = c.num_spks, c.num_stages, c.feat, c.enc_channels, c.heads
        L_ = c.frames(T)
        Lp = c.padded_frames(L_)
        Tp = Lp >> R
        st = torch.cuda.current_stream(self.device).cuda_stream
        tape: list = []
        self._call = 0

        def next_seed():
            self._call += 1
            return (seed * 1000003 + self._call * 7919) & 0xFFFFFFFFFFFFFFFF

        # encoder + projector + pad
        enc, cur, ctx = self.front_fwd(x, tp, B, T, L_, Lp)
        tape.append(("front", x, enc, ctx))

        def blk(kind, xin, w, n, Tc):
            y, rec = self.block_fwd(kind, xin, w, n, Tc, Tp, p_drop, next_seed())
            tape.append(("block", rec))
            return y

        def stage(xin, stw, n, Tc, with_spk):
            for j in range(len(stw["g"])):
                xin = blk("gcfn", blk("ega", xin, stw["g"][j][0], n, Tc), stw["g"][j][1], n, Tc)
                xin = blk("gcfn", blk("cla", xin, stw["l"][j][0], n, Tc), stw["l"][j][1], n, Tc)
                if with_spk:
                    xin = blk("gcfn", blk("spk", xin, stw["spk"][j][0], n, Tc), stw["spk"][j][1], n, Tc)
            return xin

        # temporal contracting part                                      (module.py:199-205)
        skips = []
        Tc = Lp
        for i in range(R):
            stw = tp.enc_stages[i]
            cur = stage(cur, stw, B, Tc, False)
            sk, scx = self.split_fwd(cur, tp.splits[i], B, Tc)
            tape.append(("split_skip", cur, scx, tp.splits[i], Tc, i))
            skips.append((sk, Tc))
            y, cx, To = self.down_fwd(cur, stw["down"], B, Tc)
            tape.append(("down", cur, cx, stw["down"], Tc))
            cur, Tc = y, To
        if Tc != Tp:
            raise RuntimeError(f"internal: bottleneck length {Tc} != pooled length {Tp}")
        cur = stage(cur, tp.bottleneck, B, Tc, False)
        y, scx = self.split_fwd(cur, tp.splits[R], B, Tc)
        tape.append(("split", cur, scx, tp.splits[R], Tc))
        cur = y

        # temporal expanding part                                        (module.py:207-215)
        nS = B * S
        aux: List[torch.Tensor] = []
        for i in range(R):
            if with_aux:                                                 # auxiliary head of this stage's input (model.py:47-52)
                idx = self._index(Tc, L_)
                wav_a, cx = self.head_fwd(cur, tp.out_aux[i], nS, Tc, L_, idx, enc)
                aux.append(wav_a)
                tape.append(("head_aux", cur, cx, tp.out_aux[i], Tc, idx, i))
            skip, Ts = skips[R - 1 - i]
            y = self.fuse_fwd(cur, skip, tp.fuse[i], nS, Ts)
            tape.append(("fuse", cur, skip, tp.fuse[i], Ts, R - 1 - i))
            cur, Tc = y, Ts
            cur = stage(cur, tp.dec_stages[i], nS, Tc, True)
        wav, cx = self.head_fwd(cur, tp.out_main, nS, Tc, L_, None, None)
        tape.append(("head_main", cur, cx, tp.out_main, Tc))
        return wav, aux, tape, (B, T, L_, Lp, Tp)

    # ---- backward -----------------------------------------------------------------------------------------------------------
    def backward(self, tape: list, dims, d_wav: Optional[torch.Tensor], d_aux: List[Optional[torch.Tensor]], tp: TrainPack, p_drop: float,
                 on_decoder_done=None):
        """Replays the tape in reverse.  ``d_wav`` ``[S,B,T']`` / ``d_aux[i]`` gradients of the outputs (``None`` = zero).
        ``on_decoder_done()`` is called once the expanding half of the U-Net (decoder stages, fusion convs, heads) has been
        back-propagated: from then on the tail of the gradient buffer is final (``dist.GradSync.begin``)."""
        c = self.cfg
        B, T, L_, Lp, Tp = dims
        S, F, N = c.num_spks, c.feat, c.enc_channels
        nS = B * S
        Tout = (L_ - 1) * c.enc_stride + c.enc_kernel
        # Deferred gradient finishers (include/sepr.h sepr_train_defer_begin): the ~320 weight-sized finisher launches of the walk below are
        # queued by the library and run as ~45 batched launches - at the early all-reduce point (the decoder half's gradients must be final
        # there) and at the end.  The arena holds the reduced contractions in between: about one model's worth of parameters.
        if not self._defer:
            return self._backward_walk(tape, dims, d_wav, d_aux, tp, p_drop, on_decoder_done)
        st_main = torch.cuda.current_stream(self.device).cuda_stream
        if self._wg_on:
            if self._wg_side is None:
                self._wg_side = torch.cuda.Stream(device=self.device)
            L.check(self.lib.sepr_train_wgrad_stream(self._wg_side.cuda_stream), "sepr_train_wgrad_stream")
            self._wg_active = True
            self._ws_slot = 0
        try:
            self._backward_deferred(tape, dims, d_wav, d_aux, tp, p_drop, on_decoder_done)
        finally:
            if self._wg_active:
                self._wg_active = False
                self.lib.sepr_train_wgrad_join(st_main)
                self.lib.sepr_train_wgrad_stream(None)
                self._wg_keep.clear()

    def _backward_deferred(self, tape, dims, d_wav, d_aux, tp, p_drop, on_decoder_done):
        if self._fin_arena is None:
            n_par = sum(int(v.numel()) for v in tp.sd.values() if v.dtype == torch.float32)
            self._fin_arena = torch.empty(int(1.25 * 4 * n_par) + (8 << 20), dtype=torch.uint8, device=self.device)
        st_ = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.sepr_train_defer_begin(self._fin_arena.data_ptr(), self._fin_arena.numel()), "sepr_train_defer_begin")
        if self._ride and not self._wg_on:
            # riding reductions (include/sepr.h sepr_train_defer_parts): the partial tiles of a contraction wait in one half of this double
            # buffer for the next contraction's launch, which runs their reduction in appended blocks (~290 launches fewer per step)
            if self._parts is None:
                self._parts = torch.empty(2 * (40 << 20), dtype=torch.uint8, device=self.device)
            L.check(self.lib.sepr_train_defer_parts(self._parts.data_ptr(), self._parts.numel()), "sepr_train_defer_parts")
        try:
            self._backward_walk(tape, dims, d_wav, d_aux, tp, p_drop, on_decoder_done)
        except BaseException:
            self.lib.sepr_train_defer_flush(1, st_)       # close the window, but let the walk's own exception through
            raise
        L.check(self.lib.sepr_train_defer_flush(1, st_), "sepr_train_defer_flush")

    def _backward_walk(self, tape: list, dims, d_wav, d_aux, tp: TrainPack, p_drop: float, on_decoder_done=None):
        c = self.cfg
        B, T, L_, Lp, Tp = dims
        S, F, N = c.num_spks, c.feat, c.enc_channels
        nS = B * S
        Tout = (L_ - 1) * c.enc_stride + c.enc_kernel
        dcur: Optional[torch.Tensor] = None          # gradient w.r.t. the running activation
        dskips: Dict[int, torch.Tensor] = {}         # gradient w.r.t. the split skip tensors, by encoder level
        denc = torch.zeros(B, L_, N, dtype=torch.float32, device=self.device)
        any_aux = False
        enc_saved = tape[0][2]                       # the auxiliary heads mask with the encoder output ("front" record)
        for rec in reversed(tape):
            kind = rec[0]
            if self._wg_active and dcur is not None:
                # a side-stream contraction may read the gradient tensor a block was given after the walk has dropped it: hold the last few
                self._wg_keep.append(dcur)
                if len(self._wg_keep) > 6:
                    del self._wg_keep[0]
            if kind == "block":
                dcur = self.block_bwd(rec[1], dcur)
            elif kind == "head_main":
                _, xin, cx, w, Tc = rec
                dcur = self._new(nS, Tc, F)
                if d_wav is None:
                    dcur.zero_()
                else:
                    self.head_bwd(xin, cx, w, d_wav.contiguous(), dcur, False, None, nS, Tc, L_, None, None)
            elif kind == "head_aux":
                _, xin, cx, w, Tc, idx, i = rec
                g = d_aux[i] if i < len(d_aux) else None
                if g is not None:
                    any_aux = True
                    if g.shape[-1] != Tout or not g.is_contiguous():      # Model.forward crops the aux outputs (model.py:51)
                        gp_ = torch.zeros(S, B, Tout, dtype=torch.float32, device=self.device)
                        gp_[..., : g.shape[-1]] = g
                        g = gp_
                    self.head_bwd(xin, cx, w, g, dcur, True, denc, nS, Tc, L_, idx, enc_saved)
            elif kind == "fuse":
                _, lo, skip, w, Ts, level = rec
                dcur, dskips[level] = self.fuse_bwd(lo, skip, w, dcur, nS, Ts)
            elif kind == "split":
                _, xin, cx, w, Tc = rec
                if on_decoder_done is not None:
                    # (the decoder half's queued finishers run now: the early all-reduce bucket reads their gradients; a no-op without a window)
                    L.check(self.lib.sepr_train_defer_flush(0, torch.cuda.current_stream(self.device).cuda_stream), "sepr_train_defer_flush")
                    on_decoder_done()
                dx = torch.empty_like(xin)
                self.split_bwd(xin, cx, w, dcur, dx, False, B, Tc)
                dcur = dx
            elif kind == "down":
                _, xin, cx, w, Tc = rec
                dcur = self.down_bwd(xin, cx, w, dcur, B, Tc)
            elif kind == "split_skip":
                _, xin, cx, w, Tc, level = rec
                # dcur holds d(stage output) from the DownConv; the skip path adds to it
                self.split_bwd(xin, cx, w, dskips.pop(level), dcur, True, B, Tc)
            elif kind == "front":
                _, xin, enc, cx = rec
                self.front_bwd(xin, enc, cx, tp, dcur, denc if any_aux else None, B, T, Lp)
            else:
                raise RuntimeError(f"unknown tape record {kind}")


def _fuse_fwd_w(tw: L.FuseTW) -> L.FuseW:
    """The fusion conv's forward is the inference entry point ``sepr_fuse_fwd``; it takes the (w, b, x3) triple."""
    return L.FuseW(w=tw.l.w, b=tw.l.b, x3=L.X3W(wp=tw.l.wp, bias=tw.l.b))
