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
C ABI).  There is no CPU implementation in this package; a CPU tensor raises.

``train()`` mode (reference ``engine.py:50-83``): the forward goes through ``TrainEngine`` - batch-statistics
BatchNorm with running-stat updates, dropout from a counter-based generator - inside ONE ``torch.autograd.Function``
(``_SeparatorFn``): PyTorch is the autograd glue between the criterion and the parameters' ``.grad``, the backward of
every block is an explicit ``sepr_*_bwd`` call, no per-op autograd graph exists.
"""
from __future__ import annotations

import itertools
import os
import threading
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from .config import SepConfig
from .engine import SeparatorEngine
from .pack import PackedModel
from .train_pack import PRECISIONS
from .params import KINDS, build_param_tree, mutation_epoch


DEFAULT_PRECISION = "bf16x3"

# (device index, precision, weights key, id of the owning module) -> PackedModel; see Model._packed
_PACK_CACHE: Dict[tuple, PackedModel] = {}
_PACK_LOCK = threading.Lock()
_OPTIMISTIC = os.environ.get("SEPR_OPTIMISTIC", "1") != "0"      # Model.forward: enqueue first, verify the weights identity behind it
_UIDS = itertools.count(1)


def _evict_packed(uid: int) -> None:
    with _PACK_LOCK:
        for k in [k for k in _PACK_CACHE if k[3] == uid]:
            del _PACK_CACHE[k]


class _TrainGraph:
    """One captured training step for one (batch, samples) shape: the per-step weight re-pack AND the ~2 500 launches of the
    train-mode forward are one hipGraph, the ~2 700 launches of the backward another.  A step is then two graph replays plus
    the (eager) criteria and optimizer: host time per step drops from ~100 ms of Python / ctypes / launch calls to a few ms.

    Static state: the input buffer, the outputs, every saved-for-backward context (allocated inside the capture, so it lives in
    the graphs' private pool), the flat gradient buffer, the output-gradient buffers.  Dropout: the by-value seeds are frozen
    into the graph; the per-step seed goes into the device word ``TrainPack.salt`` that every dropout kernel XORs in
    (``include/sepr.h`` seed_salt).  BatchNorm running statistics and ``num_batches_tracked`` are updated by the replay itself.
    The graph reads the parameters where they live (optimizer steps are in place); anything that moves them (``.to()``,
    re-bound parameters) changes ``Model._graph_key`` and triggers a re-capture."""

    CAPTURE_SEED = 0x5EED5EED5EED

    def __init__(self, model: "Model", x: torch.Tensor):
        from .train_engine import TrainEngine
        from .train_pack import GradBuffer, TrainPack
        dev = x.device
        self.model_ref = weakref.ref(model)
        cfg = model.cfg
        eng = model.__dict__.get("_train_engine")
        if eng is None or eng.device != dev:
            eng = TrainEngine(cfg, dev)
            model.__dict__["_train_engine"] = eng
        self.eng = eng
        self.p = float(model.dropout_p)
        self.with_aux = bool(model.compute_aux)
        names = list(model._kinds)
        flat = model._flat_tensors()
        sd = {n: t.detach() for n, t in zip(names, flat)}
        self.x = x.detach().to(torch.float32).contiguous().clone()
        # ---- warm-up (eager): fills the engine's size / index caches and grows its workspace to this shape, so that the capture
        #      contains launches only.  It is a real train-mode forward: BatchNorm state is restored afterwards.
        keep = {n: t.detach().clone() for n, t, kind in zip(names, flat, model._kinds.values()) if KINDS[kind]}
        gb0 = GradBuffer(cfg, dev)
        tp0 = TrainPack(cfg, sd, gb0, model.precision)
        wav, aux, tape, dims = eng.forward(self.x, tp0, self.p, self.CAPTURE_SEED, with_aux=self.with_aux)
        eng.backward(tape, dims, torch.zeros_like(wav), [torch.zeros_like(a) for a in aux], tp0, self.p)
        for n, t in keep.items():
            sd[n].copy_(t)
        del tp0, gb0, wav, aux, tape
        torch.cuda.synchronize(dev)
        # ---- capture
        self.gb = GradBuffer(cfg, dev)
        self.salt = torch.zeros(1, dtype=torch.int64, device=dev)     # outside the capture: replays must not reset it
        self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.g_fwd, pool=pool, capture_error_mode="thread_local"):    # (see train_step.CAPTURE_MODE)
            self.tp = TrainPack(cfg, sd, self.gb, model.precision, salt=self.salt)
            self.wav, self.aux, self.tape, self.dims = eng.forward(self.x, self.tp, self.p, self.CAPTURE_SEED, with_aux=self.with_aux)
            torch._foreach_add_(self.tp.bn_counters, 1)          # BatchNorm.num_batches_tracked
        self.d_wav = torch.zeros_like(self.wav)
        self.d_aux = [torch.zeros_like(a) for a in self.aux]
        with torch.cuda.graph(self.g_bwd, pool=pool, capture_error_mode="thread_local"):
            self.gb.flat.zero_()
            eng.backward(self.tape, self.dims, self.d_wav, self.d_aux, self.tp, self.p)
        # the capture itself executed nothing: BatchNorm state is still the pre-warm-up one
        self.replays = 0
        # the captured launches carry raw pointers into the engine's scratch workspace: hold the tensor, so that a later, larger
        # shape (which makes the engine allocate a bigger workspace) cannot hand this memory back to the allocator under the graph
        self.ws = eng._ws

    def forward(self, x: torch.Tensor, seed: int):
        self.x.copy_(x)
        self.salt.fill_(seed & 0x7FFFFFFFFFFFFFFF)
        self.g_fwd.replay()
        self.replays += 1
        # fresh tensor objects over the static buffers (autograd attaches a node to what a Function returns)
        return (self.wav.detach(), *[a.detach() for a in self.aux])

    def backward(self, d_wav, d_aux):
        if d_wav is None:
            self.d_wav.zero_()
        else:
            self.d_wav.copy_(d_wav)
        for buf, g in zip(self.d_aux, list(d_aux) + [None] * (len(self.d_aux) - len(d_aux))):
            if g is None:
                buf.zero_()
            elif g.shape == buf.shape:
                buf.copy_(g)
            else:                                                  # Model.forward crops the aux outputs (model.py:51)
                buf.zero_()
                buf[..., : g.shape[-1]].copy_(g)
        self.g_bwd.replay()
        # one copy out of the static buffer: the parameters' .grad must not alias memory the next replay rewrites
        return self.gb.flat.clone()


class _SeparatorFn(torch.autograd.Function):
    """Whole-model forward / backward through the HIP training path.  Inputs: the module, the mixture, then every
    parameter (so autograd routes the returned gradients into ``.grad``); outputs: main waveforms ``[S,B,T']`` and the R
    auxiliary ones."""

    @staticmethod
    def forward(ctx, model, x, *params):
        from .train_engine import TrainEngine
        from .train_pack import GradBuffer, TrainPack
        dev = x.device
        if model.train_graphs:
            with torch.cuda.device(dev):
                tg = model._train_graph(x)
                seed = model._next_dropout_seed() if model.dropout_p > 0.0 else 0
                outs = tg.forward(x.detach().to(torch.float32), seed)
            model.invalidate_packed()
            ctx.state = ("graph", model, tg, tg.replays)
            return outs
        with torch.cuda.device(dev):
            eng = model.__dict__.get("_train_engine")
            if eng is None or eng.device != dev:
                eng = TrainEngine(model.cfg, dev)
                model.__dict__["_train_engine"] = eng
            names = list(model._kinds)
            flat = model._flat_tensors()
            sd = {n: t.detach() for n, t in zip(names, flat)}
            gb = GradBuffer(model.cfg, dev)                  # fresh zeros per step: autograd may keep views of it as .grad
            salt = model.dropout_salt if (model.dropout_salt is not None and model.dropout_salt.device == dev) else None
            tp = TrainPack(model.cfg, sd, gb, model.precision, salt=salt)
            seed = model._next_dropout_seed() if model.dropout_p > 0.0 else 0
            wav, aux, tape, dims = eng.forward(x.detach().to(torch.float32), tp, model.dropout_p, seed, with_aux=model.compute_aux)
            torch._foreach_add_(tp.bn_counters, 1)           # BatchNorm.num_batches_tracked
        model.invalidate_packed()                            # running statistics changed behind the version counters
        ctx.state = (model, eng, tp, gb, tape, dims, len(aux))
        return (wav, *aux)

    @staticmethod
    def backward(ctx, d_wav, *d_aux):
        if ctx.state is None:
            raise RuntimeError("the HIP training path keeps one tape per forward: backward twice needs a second forward")
        if ctx.state[0] == "graph":
            _, model, tg, replay_no = ctx.state
            ctx.state = None
            if tg.replays != replay_no:
                # one static tape per shape: a second same-shape forward has overwritten the activations this backward needs
                raise RuntimeError("train_graphs mode keeps ONE static tape per input shape: another forward of the same shape ran before this "
                                   "backward (e.g. (model(x1) + model(x2)).backward()); use the eager train path (train_graphs = False) for that")
            with torch.cuda.device(tg.x.device):
                flat = tg.backward(d_wav, d_aux)
                model.__dict__["_grad_flat"] = flat               # (optim.FlatAdamW, train_step.py)
                if model.grad_sync is not None:
                    model.grad_sync(flat)                         # (no early bucket: the backward is one graph)
            grads = []
            for name, kind in model._kinds.items():
                if KINDS[kind]:
                    continue
                off, shape = tg.gb.offsets[name]
                n = 1
                for d in shape:
                    n *= d
                grads.append(flat[off:off + n].view(shape))
            return (None, None, *grads)
        model, eng, tp, gb, tape, dims, n_aux = ctx.state
        ctx.state = None
        with torch.cuda.device(eng.device):
            sync = model.grad_sync
            early = None
            if sync is not None and hasattr(sync, "begin"):
                # parameter order puts the decoder half (fusion convs, decoder stages, heads) at the tail of the buffer
                tail = gb.offsets["separator.simple_fusion.0.weight"][0]
                early = lambda: sync.begin(gb.flat, tail)          # noqa: E731
            eng.backward(tape, dims, d_wav, list(d_aux), tp, model.dropout_p, on_decoder_done=early)
            if sync is not None:
                sync(gb.flat)
            model.__dict__["_grad_flat"] = gb.flat                 # the buffer the returned gradients are views of (train_step.py)
        grads = []
        for name, kind in model._kinds.items():
            if KINDS[kind]:
                continue
            grads.append(gb.view(name))
        return (None, None, *grads)


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
        self._pack_epoch = 0
        # replicas made by torch.nn.parallel.replicate share this list (shallow __dict__ copy) and reach the original
        # module - whose parameters carry the version counters - through it
        self._origin = [weakref.ref(self)]
        self._uid = next(_UIDS)                      # cache key component (ids are recycled, uids are not)
        weakref.finalize(self, _evict_packed, self._uid)
        self.compute_aux = True    # the reference evaluates the aux heads in every forward (model.py:47-52)
        # latency mode: replay the forward from a captured hipGraph per input shape (engine.forward_graphed); the
        # returned tensors are then static buffers that the next same-shape call overwrites
        self.use_graphs = os.environ.get("SEPR_GRAPHS", "0") == "1"
        # throughput mode for batches: the batch as N independent pipelines on N streams (engine.forward_split; bit-identical
        # results).  0 = auto: two pipelines from 16 utterances up (+2.5 ... +6 % at batch 32), one below; SEPR_PIPELINES overrides.
        self.pipelines = int(os.environ.get("SEPR_PIPELINES", "0") or 0)
        # projection arithmetic: "fp32" = exact f32 MFMA; "bf16x3" = split-fp32 on the bf16 MFMA (3 MFMAs per
        # product, ~100 dB agreement with fp32, 5x less matrix time); "bf16" = plain bf16 operands with fp32 accumulation
        # and fp32 master weights - a TRAINING precision (BASELINE configs[4]); eval() forwards of a "bf16" model run in
        # bf16x3 (plain bf16 operands do not pass the 1e-3 dB SI-SNR gate).  Default from SEPR_PRECISION.
        self.precision = precision or os.environ.get("SEPR_PRECISION", DEFAULT_PRECISION)
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        # train mode: dropout probability of the GCFN / CLA sites (configs.yaml dropout_rate; 0 disables exactly) and an
        # optional callable applied to the flat gradient buffer at the end of backward (dist.GradSync: RCCL all-reduce)
        self.dropout_p = float(self.cfg.dropout)
        self.grad_sync = None
        # train mode: replay the step from captured hipGraphs (one capture per input shape; _TrainGraph).  Opt-in: the captured
        # step keeps its activations resident between steps and re-captures when the shape changes.
        self.train_graphs = os.environ.get("SEPR_TRAIN_GRAPHS", "0") == "1"
        # device word XOR-ed into every dropout seed of the EAGER train path (int64[1] or None): what lets a caller capture
        # whole steps into a hipGraph (train_step.CapturedTrainStep) and still draw fresh masks per replay
        self.dropout_salt = None

    def _graph_key(self, x: torch.Tensor):
        flat = self._flat_tensors()
        return (tuple(x.shape), x.device.index, self.precision, float(self.dropout_p), bool(self.compute_aux), len(flat),
                sum(t.data_ptr() for t in flat), mutation_epoch())

    def _train_graph(self, x: torch.Tensor) -> "_TrainGraph":
        graphs = self.__dict__.setdefault("_train_graphs", {})
        key = self._graph_key(x)
        tg = graphs.get(key)
        if tg is None:
            for k in [k for k in graphs if k[0] == key[0] or len(graphs) >= 2]:      # same shape with stale weights / precision; cap
                del graphs[k]
            tg = graphs[key] = _TrainGraph(self, x)
        return tg

    def _next_dropout_seed(self) -> int:
        """Per-step dropout seed from a PRIVATE CPU generator (the user's global RNG stream - shuffling, augmentation - is
        not consumed, as with the reference's device-side dropout), seeded from ``torch.initial_seed()`` and the
        data-parallel rank so that replicas under one ``torch.manual_seed`` draw different masks."""
        gen = self.__dict__.get("_drop_gen")
        if gen is None:
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            gen = torch.Generator()
            gen.manual_seed((torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0x7FFFFFFFFFFFFFFF)
            self.__dict__["_drop_gen"] = gen
        return int(torch.randint(0, 2 ** 62, (1,), generator=gen).item())

    # ---- weights -----------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, cfg: SepConfig, **kw) -> "Model":
        return cls(**cfg.model_kwargs(), per_level_split=cfg.per_level_split, **kw)

    def load_synthetic_(self, seed: int = 0) -> "Model":
        """Overwrite all tensors with the name-keyed O(1) synthetic weights (``synth.synth_state_dict``)."""
        from .synth import synth_state_dict
        self.load_state_dict(synth_state_dict(self.cfg, seed), strict=True)
        return self

    @property
    def infer_precision(self) -> str:
        return "bf16x3" if self.precision == "bf16" else self.precision

    # ---- packed-weight cache ------------------------------------------------------------------------
    def _flat_tensors(self) -> List[torch.Tensor]:
        """Every tensor of the state_dict, in ``param_rows`` order, resolved by attribute walk - works on the
        replicas ``torch.nn.parallel.replicate`` builds too (their parameters are plain attributes, not
        ``_parameters`` entries, so ``state_dict()`` / ``parameters()`` do not see them)."""
        cached = None if self._is_replica_module() else self.__dict__.get("_flat")
        if cached is not None and cached[0] == mutation_epoch():
            return cached[1]
        flat = []
        for name in self._kinds:
            node = self
            for part in name.split("."):
                node = getattr(node, part)
            flat.append(node)
        if not self._is_replica_module():
            # valid until a tensor attribute of the tree is rebound (params.ParamNode.__setattr__ bumps the epoch)
            self.__dict__["_flat"] = (mutation_epoch(), flat)
        return flat

    def _is_replica_module(self) -> bool:
        return bool(getattr(self, "_is_replica", False))

    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .float(): storages move
        self.__dict__.pop("_flat", None)
        return super()._apply(fn, *a, **k)

    def invalidate_packed(self) -> None:
        """Force a re-pack on the next forward.  Needed only after mutations the version counters cannot see
        (``p.data.copy_()``, ``p.data.mul_()``).  Re-bound ``nn.Parameter`` objects and re-assigned ``.data`` are seen."""
        self._pack_epoch += 1

    def _weights_key(self):
        """Cheap identity of the current weights: packed copies are caches keyed by it.  In-place updates
        (optimizer steps, ``load_state_dict``, ``p.add_()``) bump a tensor's ``_version`` so the sum changes;
        ``.to()``, ``p.data = ...`` and re-bound parameters change a storage address, and every tensor's address is in the
        key (their sum); re-bound parameter OBJECTS additionally invalidate the cached object list
        (``params.mutation_epoch``).  ~0.3 ms for Base (1390 tensors) against 2.8 ms for materialising the state_dict;
        only ``p.data.copy_()``-style writes are invisible to it - call ``invalidate_packed()`` after those."""
        flat = self._flat_tensors()
        return (len(flat), sum(t.data_ptr() for t in flat), sum(t._version for t in flat), self._pack_epoch)

    def _packed(self, dev: torch.device) -> PackedModel:
        """Packed weights for ``dev``, shared process-wide: keyed by (device, precision, identity of the ORIGINAL
        module's weights), so the replicas that ``torch.nn.parallel.data_parallel`` rebuilds on every forward
        (reference engine.py:64,98,130,167 with several device ids) hit the copy packed by an earlier replica on the
        same device instead of re-packing ~750 tensors per call."""
        origin = self._origin[0]() if self._is_replica_module() else self
        wkey = (origin if origin is not None else self)._weights_key()
        self.__dict__["_last_wkey"] = (wkey, self.infer_precision)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), self.infer_precision, wkey, self._uid)
        with _PACK_LOCK:
            pk = _PACK_CACHE.get(key)
            if pk is not None:
                return pk
        names = list(self._kinds)
        sd = {n: t.detach() for n, t in zip(names, self._flat_tensors())}
        with torch.cuda.device(dev):
            pk = PackedModel(self.cfg, sd, self.infer_precision)
        with _PACK_LOCK:
            stale = [k for k in _PACK_CACHE if k[0] == key[0] and k[3] == key[3] and k[1] == key[1] and k != key]
            for k in stale:                              # older weight versions of the same module on this device
                del _PACK_CACHE[k]
            _PACK_CACHE[key] = pk
        return pk

    def engine(self, device: Optional[torch.device] = None) -> SeparatorEngine:
        dev = device if device is not None else self._flat_tensors()[0].device
        if dev.type != "cuda":
            raise RuntimeError(
                "sepreformer_amd.Model computes on an MI355X (HIP) device only; move the module with "
                ".to('cuda'). There is deliberately no CPU fallback (the CPU restatement lives in oracle/ "
                "and is test infrastructure).")
        pk = self._packed(dev)
        if self._is_replica_module():
            # replicas are throw-away objects driven by one Python thread each (parallel_apply): a private engine
            # (workspace, side stream) per call keeps the threads independent
            with torch.cuda.device(dev):
                return SeparatorEngine(self.cfg, pk, dev)
        if self._engine is None or self._engine.pk is not pk:
            with torch.cuda.device(dev):
                self._engine = SeparatorEngine(self.cfg, pk, dev)
        self.__dict__["_engine_wkey"] = self.__dict__.get("_last_wkey")      # the weights identity self._engine was packed from
        return self._engine

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if x.dim() == 1:
            # the reference advertises [T] input but fails in GroupNorm for it
            raise RuntimeError("Expected [batch, samples] input")
        if not x.is_cuda:
            raise RuntimeError("input tensor is not on the HIP device (no CPU fallback exists)")
        if self.training:
            return self._forward_train(x)

        def run(eng):
            with torch.cuda.device(x.device):
                if self.use_graphs:
                    return eng.forward_graphed(x.to(torch.float32), with_aux=self.compute_aux)
                if self.effective_pipelines(x.shape[0]) > 1:
                    return eng.forward_split(x.to(torch.float32), with_aux=self.compute_aux, parts=self.effective_pipelines(x.shape[0]))
                return eng.forward(x.to(torch.float32), with_aux=self.compute_aux)

        # Single-utterance latency: checking the identity of 1390 weight tensors (_weights_key, ~0.3 ms) BEFORE the first launch is ~8 % of a
        # batch-1 forward.  With an engine from an earlier call the forward is enqueued first, from its packed weights, and the identity is
        # checked while the device works; the (rare) mismatch discards that result and takes the regular path below - nothing computed from
        # stale weights is ever returned.  (SEPR_OPTIMISTIC=0 restores the check-first order.)
        eng0 = self._engine
        if (_OPTIMISTIC and eng0 is not None and not self._is_replica_module() and eng0.device == x.device
                and self.__dict__.get("_engine_wkey") is not None):
            out0 = run(eng0)
            if (self._weights_key(), self.infer_precision) == self.__dict__.get("_engine_wkey"):
                wav, aux = out0
                T = x.shape[-1]
                return [wav[s] for s in range(self.num_spks)], [[a[s][..., :T] for s in range(self.num_spks)] for a in aux]
            del out0
        eng = self.engine(x.device if self._is_replica_module() else None)
        wav, aux = run(eng)
        T = x.shape[-1]
        audio = [wav[s] for s in range(self.num_spks)]
        audio_aux = [[a[s][..., :T] for s in range(self.num_spks)] for a in aux]
        return audio, audio_aux

    def effective_pipelines(self, batch: int) -> int:
        """Number of sub-batch pipelines ``forward`` uses for ``batch`` utterances in eval mode (``pipelines`` = 0: auto)."""
        pl = self.pipelines if self.pipelines > 0 else (2 if batch >= 16 and not self._is_replica_module() else 1)
        return pl if (pl > 1 and batch >= 8 * pl) else 1

    def _forward_train(self, x: torch.Tensor):
        """Reference ``Model.forward`` under ``model.train()`` (engine.py:51,64): same return structure, autograd-connected."""
        params = [t for t, kind in zip(self._flat_tensors(), self._kinds.values()) if not KINDS[kind]]
        outs = _SeparatorFn.apply(self, x, *params)
        wav, aux = outs[0], outs[1:]
        T = x.shape[-1]
        audio = [wav[s] for s in range(self.num_spks)]
        audio_aux = [[a[s][..., :T] for s in range(self.num_spks)] for a in aux]
        return audio, audio_aux

    @torch.no_grad()
    def separate(self, x: torch.Tensor) -> torch.Tensor:
        """Inference convenience: main outputs only, ``[S,B,T']`` in one tensor."""
        wav, _ = self.engine().forward(x, with_aux=False)
        return wav
