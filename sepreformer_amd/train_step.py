"""A whole training step as hipGraph replays: forward + criteria + backward in one captured graph, gradient clipping + optimizer
step in a second one, the data-parallel gradient all-reduce (RCCL) between them.

Why: the reference's step (``engine.py:60-77``: forward, three criteria calls, ``loss.backward()``, ``clip_grad_norm_``,
``optimizer.step()``) is ~4 200 kernel launches here.  ``Model.train_graphs`` already replays the separator's forward and backward,
but criteria, clipping and the optimizer stayed eager: ~200 launches the host issues AFTER the backward graph has been submitted,
and a kernel trace of the replayed step shows the device idle for 5-10 ms per step waiting for them.  Capturing the whole step
leaves three host calls per step (replay, all-reduce, replay).

    step = CapturedTrainStep(model, loss_fn, optimizer, x, targets, max_norm=5.0)
    for x, targets in loader:                      # same shapes as the example batch
        loss, grad_norm = step(x, targets)         # device tensors, overwritten by the next call

``loss_fn(audio, aux, *targets) -> scalar`` is the caller's (e.g. the reference's 0.6 / 0.4 mix of ``PIT_SISNR_time`` and
``PIT_SISNR_mag``); the optimizer must be capturable (``torch.optim.AdamW(..., capturable=True)``: its step counter lives on the
device).  The ``warmup`` eager iterations the capture needs (workspace growth, optimizer state) are REAL training steps on the
example batch.  Dropout: the by-value seeds frozen into the graph are XOR-ed with ``model.dropout_salt`` on the device, which
every call refreshes from the model's seed stream (``include/sepr.h`` seed_salt), so each replay draws fresh masks.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

# "thread_local": calls other threads make while this one captures are theirs to answer for - the RCCL watchdog thread of an
# initialised process group polls its events with hipEventQuery, which the default "global" mode turns into a capture error
# (and a dead process) whenever a collective ran shortly before the capture
CAPTURE_MODE = "thread_local"


class CapturedTrainStep:
    def __init__(self, model, loss_fn: Callable, optimizer: torch.optim.Optimizer, example_x: torch.Tensor,
                 example_targets: Sequence[torch.Tensor], max_norm: Optional[float] = None, warmup: int = 2):
        if example_x.device.type != "cuda":
            raise RuntimeError("CapturedTrainStep needs the HIP device (no CPU path exists)")
        if not all(bool(g.get("capturable", False)) for g in optimizer.param_groups):
            raise ValueError("the optimizer must be built with capturable=True (its step counter has to live on the device)")
        if warmup < 1:
            raise ValueError("at least one eager warm-up step is needed (optimizer state, workspaces)")
        dev = example_x.device
        self.model, self.loss_fn, self.opt, self.max_norm = model, loss_fn, optimizer, max_norm
        self.params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self.x = example_x.detach().clone()
        self.targets = [t.detach().clone() for t in example_targets]
        self._graphs_before = bool(model.train_graphs)
        model.train()
        model.train_graphs = False                   # the captured region runs the eager path (graphs do not nest)
        if model.dropout_salt is None or model.dropout_salt.device != dev:
            model.dropout_salt = torch.zeros(1, dtype=torch.int64, device=dev)
        self.salt = model.dropout_salt
        self.sync = model.grad_sync
        self.calls = 0

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._refresh_salt()
                optimizer.zero_grad(set_to_none=True)
                audio, aux = model(self.x)
                loss = loss_fn(audio, aux, *self.targets)
                loss.backward()
                self._clip_and_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if hasattr(optimizer, "validate"):
            optimizer.validate()                     # (optim.FlatAdamW: every pointer of its tables, once, before they are baked into a graph)
        if self.sync is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            # The warm-up steps ran gradient all-reduces.  Their completion events are polled by the process group's watchdog thread
            # (hipEventQuery, every 100 ms) until it has retired them, and HIP refuses such a query for an event whose stream is capturing -
            # the watchdog then throws and the process aborts (round 4's ~1-in-15 SIGABRT; reproduced with a backtrace in round 5: DESIGN.md
            # section 10c).  dist.all_reduce_off_stream keeps those events off every stream a capture can touch; this wait (two polling
            # periods after everything has completed) additionally lets the watchdog empty its list before the capture starts, which also
            # covers collectives issued by the caller's own code with plain dist.all_reduce.
            import os
            import time
            time.sleep(float(os.environ.get("SEPR_CAPTURE_DRAIN_S", "0.25") or 0.0))       # (0 = the round-4 behaviour, for tools/rccl_watchdog_loop.sh)

        self.g_main, self.g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        optimizer.zero_grad(set_to_none=True)
        model.grad_sync = None                       # the all-reduce runs between the two graphs, outside any capture
        try:
            with torch.cuda.graph(self.g_main, pool=pool, capture_error_mode=CAPTURE_MODE):
                audio, aux = model(self.x)
                self.loss = loss_fn(audio, aux, *self.targets)
                self.loss.backward()
        finally:
            model.grad_sync = self.sync
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("a parameter of the optimizer received no gradient during the capture")
        # the gradients autograd installed must alias the flat buffer of the captured backward: the all-reduce between the two
        # graphs works on that buffer and the optimizer graph reads the parameters' .grad
        flat = model.__dict__.get("_grad_flat")
        lo = flat.data_ptr() if flat is not None else 0
        hi = lo + (flat.numel() * flat.element_size() if flat is not None else 0)
        aliased = flat is not None and all(lo <= g.data_ptr() and g.data_ptr() + g.numel() * g.element_size() <= hi for g in grads)
        self.flat = flat if aliased else None
        if self.sync is not None and self.flat is None:
            raise RuntimeError("the parameters' .grad do not alias the flat gradient buffer: cannot all-reduce between the graphs")
        with torch.cuda.graph(self.g_opt, pool=pool, capture_error_mode=CAPTURE_MODE):
            self.grad_norm = self._clip_and_step()
        # the captured launches point into the train engine's scratch workspace: keep that tensor alive for the graphs' lifetime
        # (a later, larger eager shape makes the engine allocate a new one; the old one must not go back to the allocator)
        eng = model.__dict__.get("_train_engine")
        self._ws_keep = eng._ws if eng is not None else None
        self.audio, self.aux = audio, aux            # static outputs of the last replay (detached views are the caller's business)

    def _clip_and_step(self):
        """engine.py:76-77.  An optimizer with ``fused_clip`` (optim.FlatAdamW) clips inside its step."""
        if getattr(self.opt, "fused_clip", False):
            return self.opt.step(max_norm=self.max_norm)
        gn = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm) if self.max_norm is not None else None
        self.opt.step()
        return gn

    def _refresh_salt(self):
        if self.model.dropout_p > 0.0:
            self.salt.fill_(self.model._next_dropout_seed() & 0x7FFFFFFFFFFFFFFF)

    def __call__(self, x: torch.Tensor, targets: Sequence[torch.Tensor]):
        if x.shape != self.x.shape or len(targets) != len(self.targets):
            raise ValueError(f"captured for input {tuple(self.x.shape)} and {len(self.targets)} targets")
        self.x.copy_(x)
        for dst, src in zip(self.targets, targets):
            dst.copy_(src)
        self._refresh_salt()
        self.g_main.replay()
        if self.sync is not None:
            self.sync(self.flat)
        if hasattr(self.opt, "refresh"):
            self.opt.refresh()                       # a scheduler's new learning rate reaches the device scalar the graph reads
        self.g_opt.replay()
        self.model.invalidate_packed()               # the weights and BatchNorm state changed behind the version counters
        self.calls += 1
        return self.loss, self.grad_norm

    def release(self):
        """Drop the graphs and give the model its previous launch mode back."""
        self.model.train_graphs = self._graphs_before
        self.g_main = self.g_opt = None
