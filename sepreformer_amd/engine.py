"""Forward driver: walks the separator topology and calls one C-ABI entry point per fused block.

Topology restated from reference ``modules/module.py:190-218`` (Separator.forward) and ``model.py:38-54``
(Model.forward); every activation is a channel-last ``[sequences, frames, F]`` fp32 tensor on the HIP
device and *stays* in that layout end to end - the reference's 471 ``permute().contiguous()`` copies per
forward have no counterpart here.  PyTorch is used for the caching allocator and the
current-stream handle only.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .config import SepConfig
from .pack import GN_EPS, PackedModel


def nearest_index(src: int, dst: int) -> np.ndarray:
    """Source frame of each output frame for ``F.upsample(mode='nearest')`` (reference model.py:49).

    aten's ``nearest_idx``: identity when equal, ``>>1`` when exactly doubling, else
    ``min(floor(dst_index * float32(src/dst)), src-1)`` evaluated in fp32 (checked against torch's own
    tables in tests/golden/blocks_tiny.npz)."""
    i = np.arange(dst, dtype=np.int64)
    if dst == src:
        return i.astype(np.int32)
    if dst == 2 * src:
        return (i >> 1).astype(np.int32)
    scale = np.float32(src) / np.float32(dst)
    idx = np.floor(i.astype(np.float32) * scale).astype(np.int64)
    return np.minimum(idx, src - 1).astype(np.int32)


class SeparatorEngine:
    def __init__(self, cfg: SepConfig, packed: PackedModel, device: torch.device):
        self.cfg = cfg
        self.pk = packed
        self.device = device
        self.lib = L.load()
        self._ws: Optional[torch.Tensor] = None
        self._idx_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}
        # the speaker splits of the skip connections and the auxiliary heads depend on nothing downstream of their
        # input, so they are enqueued on a side stream and fill the CUs that the small launches of the deeper
        # stages leave idle (SEPR_OVERLAP=0: everything on one stream)
        self.overlap = os.environ.get("SEPR_OVERLAP", "1") != "0"
        self._side: Optional[torch.cuda.Stream] = None
        self._ws2: Optional[torch.Tensor] = None
        self._held: list = []
        self._peers: list = []
        self._peer_streams: list = []
        # SEPR_CHAIN_STATS=1 (round 6 experiment, OFF by default): LayerNorm statistics travel with the residual stream where no fused kernel
        # owns the LayerNorm (the generic bf16x3 projection path: F = 256) - identical results, and MEASURED NOT FASTER: the wide core's
        # statistics tail costs what the 92 removed rowstats launches saved (357 utt/s either way; profiles/r06_large_statschain_*.csv)
        self.chain_stats = (packed.precision == "bf16x3" and cfg.feat not in (64, 128) and os.environ.get("SEPR_CHAIN_STATS", "0") == "1")
        self._ys: Optional[torch.Tensor] = None       # statistics of the rows the last chained block wrote

    # ---- plumbing ---------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _workspace(self, nbytes: int) -> Tuple[int, int]:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws.data_ptr(), self._ws.numel()

    def workspace_bytes(self, B: int, L_: int, Lp: int) -> int:
        c, lib = self.cfg, self.lib
        S, F, N = c.num_spks, c.feat, c.enc_channels
        Tp = Lp >> c.num_stages
        need = [lib.sepr_workspace_bytes(L.OP_ENCODER, B, L_, 0, F, N, S),
                lib.sepr_workspace_bytes(L.OP_GCFN, B * S, Lp, 0, F, N, S),
                lib.sepr_workspace_bytes(L.OP_CLA, B * S, Lp, 0, F, N, S),
                lib.sepr_workspace_bytes(L.OP_EGA, B * S, Lp, Tp, F, N, S),
                lib.sepr_workspace_bytes(L.OP_SPKATTN, B * S, Lp, 0, F, N, S),
                lib.sepr_workspace_bytes(L.OP_SPKSPLIT, B, Lp, 0, F, N, S),
                lib.sepr_workspace_bytes(L.OP_OUTLAYER, B * S, L_, 0, F, N, S)]
        return int(max(need))

    def prepare(self, B: int, L_: int, Lp: int) -> None:
        """Bind the current stream and a workspace large enough for ``B`` utterances of ``Lp`` frames.
        ``forward`` calls this itself; tests call it before driving single blocks."""
        self._st = self._stream()
        self._wsargs = self._workspace(self.workspace_bytes(B, L_, Lp))
        self._side_on = self.overlap and not torch.cuda.is_current_stream_capturing()
        if self._side_on:
            c, lib = self.cfg, self.lib
            need = max(lib.sepr_workspace_bytes(L.OP_SPKSPLIT, B, Lp, 0, c.feat, c.enc_channels, c.num_spks),
                       lib.sepr_workspace_bytes(L.OP_OUTLAYER, B * c.num_spks, L_, 0, c.feat, c.enc_channels, c.num_spks))
            if self._ws2 is None or self._ws2.numel() < need:
                self._ws2 = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)

    def _new(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _idx(self, src: int, dst: int) -> torch.Tensor:
        key = (src, dst)
        t = self._idx_cache.get(key)
        if t is None:
            t = torch.from_numpy(nearest_index(src, dst)).to(self.device)
            self._idx_cache[key] = t
        return t

    # ---- blocks (each = one C-ABI call) --------------------------------------------------------------
    # Statistics threading (round 6, include/sepr.h "*_st"): on the generic projection path (F = 256, the Large variants) every block
    # needs the LayerNorm statistics of its input rows, which used to cost a pass over the tensor per block.  A block now hands the
    # statistics of its OUTPUT rows (computed where the last projection's tile holds whole rows) to the next block: ``xs`` in, ``ys`` out.
    # Base (fused kernels: LayerNorm inside) keeps the plain entry points.
    def _stats(self, x):
        return torch.empty(x.numel() // x.shape[-1], 2, dtype=torch.float32, device=x.device) if self.chain_stats else None

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def gcfn(self, x, w, n, T, xs=None):
        y = torch.empty_like(x)
        if not self.chain_stats:
            L.check(self.lib.sepr_gcfn_fwd(x.data_ptr(), y.data_ptr(), n, T, self.cfg.feat, C.byref(w), *self._wsargs, self._st), "sepr_gcfn_fwd")
            return y
        ys = self._stats(y)
        L.check(self.lib.sepr_gcfn_fwd_st(x.data_ptr(), self._p(xs), y.data_ptr(), ys.data_ptr(), n, T, self.cfg.feat, C.byref(w), *self._wsargs, self._st), "sepr_gcfn_fwd_st")
        self._ys = ys
        return y

    def cla(self, x, w, n, T, xs=None):
        y = torch.empty_like(x)
        if not self.chain_stats:
            L.check(self.lib.sepr_cla_fwd(x.data_ptr(), y.data_ptr(), n, T, self.cfg.feat, self.cfg.cla_kernel, C.byref(w), *self._wsargs, self._st), "sepr_cla_fwd")
            return y
        ys = self._stats(y)
        L.check(self.lib.sepr_cla_fwd_st(x.data_ptr(), self._p(xs), y.data_ptr(), ys.data_ptr(), n, T, self.cfg.feat, self.cfg.cla_kernel, C.byref(w), *self._wsargs, self._st), "sepr_cla_fwd_st")
        self._ys = ys
        return y

    def ega(self, x, w, n, T, Tp, xs=None):
        y = torch.empty_like(x)
        if not self.chain_stats:
            L.check(self.lib.sepr_ega_fwd(x.data_ptr(), y.data_ptr(), n, T, Tp, self.cfg.feat, self.cfg.heads, C.byref(w), *self._wsargs, self._st), "sepr_ega_fwd")
            return y
        ys = self._stats(y)
        L.check(self.lib.sepr_ega_fwd_st(x.data_ptr(), self._p(xs), y.data_ptr(), ys.data_ptr(), n, T, Tp, self.cfg.feat, self.cfg.heads, C.byref(w), *self._wsargs, self._st), "sepr_ega_fwd_st")
        self._ys = ys
        return y

    def spkattn(self, x, w, n, T, xs=None):
        y = torch.empty_like(x)
        if not self.chain_stats:
            L.check(self.lib.sepr_spkattn_fwd(x.data_ptr(), y.data_ptr(), n, self.cfg.num_spks, T, self.cfg.feat, self.cfg.heads, C.byref(w), *self._wsargs, self._st), "sepr_spkattn_fwd")
            return y
        ys = self._stats(y)
        L.check(self.lib.sepr_spkattn_fwd_st(x.data_ptr(), self._p(xs), y.data_ptr(), ys.data_ptr(), n, self.cfg.num_spks, T, self.cfg.feat, self.cfg.heads, C.byref(w), *self._wsargs, self._st), "sepr_spkattn_fwd_st")
        self._ys = ys
        return y

    def global_block(self, x, gw, n, T, Tp, xs=None):       # reference modules/network.py:198-209
        h = self.ega(x, gw[0], n, T, Tp, xs)
        return self.gcfn(h, gw[1], n, T, self._ys if self.chain_stats else None)

    def local_block(self, x, lw, n, T, xs=None):            # reference modules/network.py:220-224
        h = self.cla(x, lw[0], n, T, xs)
        return self.gcfn(h, lw[1], n, T, self._ys if self.chain_stats else None)

    def downconv(self, x, w, n, T):
        K = self.cfg.down_kernel
        To = (T + 2 * ((K - 1) // 2) - K) // 2 + 1
        y = self._new(n, To, self.cfg.feat)
        L.check(self.lib.sepr_downconv_fwd(x.data_ptr(), y.data_ptr(), n, T, self.cfg.feat, K, C.byref(w), self._st), "sepr_downconv_fwd")
        return y, To

    def _aside(self, fn, *inputs):
        """Run ``fn(wsargs, stream_handle)`` on the side stream once ``inputs`` (produced on the main stream) are ready.

        Lifetime rules instead of ``record_stream`` (which defers every free and made the allocator's reserved pool
        grow by gigabytes per step): the inputs are kept referenced until ``_join`` (the main stream then waits for
        the side stream, so freeing them afterwards is ordered), and the outputs live in the side stream's pool,
        whose next allocation can only happen after the next ``wait_stream(main)`` below, i.e. after every main-stream
        consumer of the previous forward has been enqueued ahead of it."""
        if not self._side_on:
            return fn(self._wsargs, self._st)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            out = fn((self._ws2.data_ptr(), self._ws2.numel()), self._side.cuda_stream)
        self._held.extend(inputs)
        return out

    def _join(self):
        if self._side_on:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._held.clear()

    def spksplit(self, x, w, B, T, wsargs=None, st=None):
        S = self.cfg.num_spks
        y = self._new(B * S, T, self.cfg.feat)
        L.check(self.lib.sepr_spksplit_fwd(x.data_ptr(), y.data_ptr(), B, S, T, self.cfg.feat, GN_EPS, C.byref(w),
                                           *(wsargs or self._wsargs), st or self._st), "sepr_spksplit_fwd")
        return y

    def fuse(self, lo, skip, wb, n, T):
        y = torch.empty_like(skip)
        L.check(self.lib.sepr_fuse_fwd(lo.data_ptr(), skip.data_ptr(), y.data_ptr(), n, T, self.cfg.feat, C.byref(wb), self._st), "sepr_fuse_fwd")
        return y

    def head(self, x, w, nS, Tsrc, L_, idx, enc, B, wsargs=None, st=None):
        c = self.cfg
        Tout = (L_ - 1) * c.enc_stride + c.enc_kernel
        wav = self._new(c.num_spks, B, Tout)
        L.check(self.lib.sepr_outlayer_decoder_fwd(
            x.data_ptr(), nS, c.num_spks, Tsrc, L_, None if idx is None else idx.data_ptr(),
            None if enc is None else enc.data_ptr(), c.feat, c.enc_channels, c.enc_kernel, c.enc_stride,
            C.byref(w), wav.data_ptr(), *(wsargs or self._wsargs), st or self._st), "sepr_outlayer_decoder_fwd")
        return wav

    # ---- two half-batch pipelines ---------------------------------------------------------------------
    @torch.no_grad()
    def forward_split(self, x: torch.Tensor, with_aux: bool = True, parts: int = 2):
        """The batch as ``parts`` independent sub-batches, each walked by its own driver thread on its own stream.

        Why: every launch runs ceil(tiles / workgroup slots) rounds, and most launches of a 32-utterance forward have
        1-4 rounds, so 20 % of the fused-GCFN time (and similar shares elsewhere) is the idle tail of a last, partly
        filled round.  Two half-size pipelines whose launches interleave on the device fill each other's tails.
        Utterances are independent and no kernel's arithmetic depends on the batch composition, so the result is
        bit-identical to ``forward``.  ctypes releases the GIL inside every launch, the threads only serialise on the
        Python bookkeeping between launches."""
        B = x.shape[0]
        if parts < 2 or B < 2 * parts or torch.cuda.is_current_stream_capturing():
            return self.forward(x, with_aux=with_aux)
        import threading
        if len(self._peers) != parts:
            self._peers = [SeparatorEngine(self.cfg, self.pk, self.device) for _ in range(parts)]
            self._peer_streams = [torch.cuda.Stream(device=self.device) for _ in range(parts)]
        main = torch.cuda.current_stream(self.device)
        chunks = list(x.contiguous().chunk(parts, 0))
        results: list = [None] * parts
        errors: list = []

        def work(i):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self._peer_streams[i]):
                    results[i] = self._peers[i].forward(chunks[i], with_aux=with_aux)
            except BaseException as e:          # re-raised on the calling thread
                errors.append(e)

        for st in self._peer_streams:
            st.wait_stream(main)
        threads = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for st in self._peer_streams:
            main.wait_stream(st)
        if errors:
            for p in self._peers:                    # peers that did not raise themselves still hold their inputs
                p._release_held()
            raise errors[0]
        wav = torch.cat([r[0] for r in results], dim=1)
        aux = [torch.cat([r[1][j] for r in results], dim=1) for j in range(len(results[0][1]))]
        return wav, aux

    # ---- latency mode: the whole forward as one hipGraph ----------------------------------------------
    @torch.no_grad()
    def forward_graphed(self, x: torch.Tensor, with_aux: bool = True):
        """Same result as ``forward`` but replayed from a captured hipGraph (one per input shape): the ~450 kernel
        launches of a forward are enqueued by the driver in one call instead of 450 Python -> ctypes -> HIP round
        trips, which is what bounds a single short utterance.  The returned tensors are the
        graph's static output buffers: they are overwritten by the next call with the same shape."""
        key = (tuple(x.shape), bool(with_aux))
        entry = self._graphs.get(key)
        if entry is None:
            static_x = x.detach().clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):              # warm-up off the capture: workspace, index tables, allocator pools
                for _ in range(2):
                    self.forward(static_x, with_aux=with_aux)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(static_x, with_aux=with_aux)
            entry = (graph, static_x, out, self._ws)   # the captured launches point into this workspace: keep it alive
            self._graphs[key] = entry
        graph, static_x, out = entry[:3]
        static_x.copy_(x)
        graph.replay()
        return out

    # ---- whole forward --------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, with_aux: bool = True, taps: Optional[dict] = None):
        """x ``[B,T]`` fp32 on the HIP device -> (wav ``[S,B,T']``, list of R aux ``[S,B,T']`` or ``[]``)."""
        try:
            return self._forward(x, with_aux, taps)
        except BaseException:
            # a launch that raised mid-forward must not leave the side-stream inputs (encoder output, stage
            # activations: hundreds of MB at B=32) referenced until the next successful forward
            self._release_held()
            raise

    def _release_held(self):
        if getattr(self, "_side_on", False) and self._side is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._held.clear()

    def _forward(self, x: torch.Tensor, with_aux: bool = True, taps: Optional[dict] = None):
        c, pk, lib = self.cfg, self.pk, self.lib
        if x.dim() != 2:
            raise RuntimeError("Input can only be 2 dimensional: [batch, samples]")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise RuntimeError("separator input must be a float32 tensor on the HIP device")
        x = x.contiguous()
        B, T = x.shape
        S, R, F, N = c.num_spks, c.num_stages, c.feat, c.enc_channels
        if T < c.enc_kernel:
            raise RuntimeError(f"input of {T} samples is shorter than the encoder kernel ({c.enc_kernel})")
        L_ = c.frames(T)
        Lp = c.padded_frames(L_)
        Tp = Lp >> R
        self.prepare(B, L_, Lp)

        # AudioEncoder + FeatureProjector + pad_signal               (model.py:39-40, module.py:193)
        enc = self._new(B, L_, N)
        gn = self._new(B, 2)
        L.check(lib.sepr_encoder_fwd(x.data_ptr(), B, T, pk.enc_w, N, c.enc_kernel, c.enc_stride, GN_EPS,
                                     enc.data_ptr(), gn.data_ptr(), *self._wsargs, self._st), "sepr_encoder_fwd")
        cur = self._new(B, Lp, F)
        L.check(lib.sepr_projector_fwd(enc.data_ptr(), B, L_, Lp, N, F, gn.data_ptr(), pk.proj_g, pk.proj_b,
                                       pk.proj_w, cur.data_ptr(), self._st), "sepr_projector_fwd")
        if taps is not None:
            taps["enc"], taps["proj"] = enc, cur

        # temporal contracting part                                   (module.py:199-205)
        skips: List[Tuple[torch.Tensor, int]] = []
        Tc = Lp
        for i in range(R):
            st = pk.enc_stages[i]
            cs = None                                                 # (the stage input comes from the projector / a DownConv: no statistics yet)
            for j in range(2):
                cur = self.global_block(cur, st["g"][j], B, Tc, Tp, cs)
                cur = self.local_block(cur, st["l"][j], B, Tc, self._ys)
                cs = self._ys
            if taps is not None:
                taps[f"enc{i}.skip_pre_split"] = cur
            skips.append((self._aside(lambda ws, sh, x_=cur, i_=i, T_=Tc: self.spksplit(x_, pk.splits[i_], B, T_, ws, sh), cur), Tc))
            cur, Tc = self.downconv(cur, st["down"], B, Tc)
        if Tc != Tp:
            raise RuntimeError(f"internal: bottleneck length {Tc} != pooled length {Tp}")
        cs = None
        for j in range(2):
            cur = self.global_block(cur, pk.bottleneck["g"][j], B, Tc, Tp, cs)
            cur = self.local_block(cur, pk.bottleneck["l"][j], B, Tc, self._ys)
            cs = self._ys
        if taps is not None:
            taps["bottleneck"] = cur
        cur = self.spksplit(cur, pk.splits[R], B, Tc)
        self._join()                                                  # the skip splits

        # temporal expanding part                                     (module.py:207-215)
        nS = B * S
        aux: List[torch.Tensor] = []
        for i in range(R):
            if with_aux:                                              # auxiliary head of this stage's input (model.py:47-52)
                aux.append(self._aside(lambda ws, sh, x_=cur, i_=i, T_=Tc: self.head(x_, pk.out_aux[i_], nS, T_, L_, self._idx(T_, L_), enc, B, ws, sh), cur, enc))
            skip, Ts = skips[R - 1 - i]
            if Ts != 2 * Tc:
                raise RuntimeError(f"internal: skip length {Ts} != 2 x {Tc}")
            cur = self.fuse(cur, skip, pk.fuse[i], nS, Ts)
            Tc = Ts
            st = pk.dec_stages[i]
            cs = None                                                 # (fusion conv output)
            for j in range(3):
                cur = self.global_block(cur, st["g"][j], nS, Tc, Tp, cs)
                cur = self.local_block(cur, st["l"][j], nS, Tc, self._ys)
                cur = self.spkattn(cur, st["spk"][j][0], nS, Tc, self._ys)
                cur = self.gcfn(cur, st["spk"][j][1], nS, Tc, self._ys)
                cs = self._ys
            if taps is not None:
                taps[f"dec{i}"] = cur
        skips.clear()

        # output layer + decoder, main and auxiliary heads            (model.py:42-52)
        wav = self.head(cur, pk.out_main, nS, Tc, L_, None, None, B)
        self._join()                                                  # the auxiliary heads
        return wav, aux
