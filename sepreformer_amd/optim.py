"""The optimizer step of the reference loop as three launches: ``clip_grad_norm_`` + ``torch.optim.AdamW.step``
(``engine.py:76-77``) over the training path's flat gradient buffer (``include/sepr.h`` sepr_adamw_step).

    opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-2)
    loss.backward()
    grad_norm = opt.step(max_norm=5.0)          # device scalar: the total norm BEFORE clipping (what clip_grad_norm_ returns)

Why: torch reaches Base's 1 312 parameter tensors through ``multi_tensor_apply`` - 37 launches for the update, 12 for the norms, 37
for the clip scaling, 1.8 ms per step - although the training path already keeps every gradient in ONE buffer.  Here the norm is one
pass over that buffer and the update one launch over a (tensor, first element) block table.

Same arithmetic as ``torch.optim.AdamW`` (decoupled weight decay, bias corrections from the device-side step counter); the
gradients are NOT scaled in place (the clip coefficient is applied inside the update).  ``param_groups[0]["lr"]`` may be changed by
a scheduler: the value is mirrored into a device scalar before each step, so a captured step (``train_step.CapturedTrainStep``)
picks it up without a re-capture.  State (``exp_avg`` / ``exp_avg_sq`` / ``step``) is exposed per parameter as views of the flat
buffers, so ``state_dict()`` has ``torch.optim.AdamW``'s layout.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L


class FlatAdamW(torch.optim.Optimizer):
    fused_clip = True          # train_step.CapturedTrainStep: step(max_norm=...) does the clipping itself

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        if len(params) != sum(1 for _ in model.parameters()):
            # the norm pass runs over the WHOLE flat gradient buffer, which holds the gradient of every parameter of the model
            raise ValueError("FlatAdamW clips over the model's whole flat gradient buffer: every parameter must be trainable")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW needs the parameters on the HIP device (no CPU path exists)")
        for p in params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16:
                raise ValueError("FlatAdamW: every parameter must be a contiguous, 16-byte aligned fp32 tensor on one device")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True))
        self._model, self._dev, self._lib = model, dev, L.load()
        self._params = params
        blk = self._lib.sepr_adamw_block_elems()
        # moments: 64-element aligned slots, like the gradient buffer
        soff, off = [], 0
        for p in params:
            soff.append(off)
            off += (p.numel() + 63) // 64 * 64
        self._exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.float64, device=dev)
        self._lr = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)
        self._scal = torch.zeros(8, dtype=torch.float32, device=dev)
        self._ws = torch.zeros(self._lib.sepr_adamw_workspace(), dtype=torch.uint8, device=dev)
        blocks = []
        for i, p in enumerate(params):
            blocks.extend((i, e) for e in range(0, p.numel(), blk))
        self._t_params = torch.tensor([p.data_ptr() for p in params], dtype=torch.int64, device=dev)
        self._t_soff = torch.tensor(soff, dtype=torch.int64, device=dev)
        self._t_numel = torch.tensor([p.numel() for p in params], dtype=torch.int32, device=dev)
        self._t_blocks = torch.tensor(blocks, dtype=torch.int32, device=dev).contiguous()
        self._nblocks = len(blocks)
        self._soff = soff
        self._t_goff: Optional[torch.Tensor] = None           # built from the first gradients (their offsets in the flat buffer)
        self._goff_host: Optional[list] = None
        self._ptrs = [p.data_ptr() for p in params]
        self._expose_state()

    # ---- torch.optim.Optimizer plumbing ---------------------------------------------------------------------------------
    def _expose_state(self):
        for p, so in zip(self._params, self._soff):
            n = p.numel()
            self.state[p] = {"step": self._step[0], "exp_avg": self._exp_avg[so:so + n].view(p.shape),
                             "exp_avg_sq": self._exp_avg_sq[so:so + n].view(p.shape)}

    def load_state_dict(self, state_dict):
        """Accepts a ``torch.optim.AdamW`` / ``FlatAdamW`` state dict: the moments are copied INTO the flat buffers."""
        if len(state_dict.get("param_groups", [])) != 1:
            raise ValueError("FlatAdamW has ONE parameter group (one set of hyper-parameters for the whole flat buffer)")
        super().load_state_dict(state_dict)
        for grp in self.param_groups:           # a torch.optim.AdamW checkpoint carries capturable=False: the step here is always capturable
            grp["capturable"] = True
        steps = []
        for p, so in zip(self._params, self._soff):
            st = self.state.get(p, {})
            if "exp_avg" in st:
                n = p.numel()
                self._exp_avg[so:so + n].copy_(st["exp_avg"].reshape(-1).to(self._dev, torch.float32))
                self._exp_avg_sq[so:so + n].copy_(st["exp_avg_sq"].reshape(-1).to(self._dev, torch.float32))
                steps.append(float(st["step"]))
        if steps:
            if max(steps) != min(steps):
                raise ValueError("FlatAdamW keeps ONE step counter: the loaded per-parameter steps differ")
            self._step.fill_(steps[0])
        self._lr_host = None
        self._expose_state()

    def add_param_group(self, param_group):
        """One group only: the update reads ``param_groups[0]``'s hyper-parameters for every tensor of the flat buffer."""
        if getattr(self, "param_groups", None):
            raise ValueError("FlatAdamW supports a single parameter group (per-group lr / weight decay would be silently ignored)")
        super().add_param_group(param_group)

    def refresh(self):
        """Mirror the (scheduler-updated) learning rate into the device scalar.  Called by ``step`` outside a capture and by
        ``CapturedTrainStep`` before each replay."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self._lr.fill_(lr)
            self._lr_host = lr

    # ---- the step -----------------------------------------------------------------------------------------------------------
    def _grad_base(self) -> torch.Tensor:
        flat = self._model.__dict__.get("_grad_flat")
        g0 = self._params[0].grad
        if flat is None or g0 is None:
            raise RuntimeError("FlatAdamW.step: no gradients of the HIP training path to apply (call loss.backward() on model(x) first)")
        if self._goff_host is None:
            lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
            offs = []
            for p in self._params:
                g = p.grad
                if g is None or g.dtype != torch.float32 or not g.is_contiguous() or not (lo <= g.data_ptr() and g.data_ptr() + 4 * g.numel() <= hi):
                    raise RuntimeError("FlatAdamW: a parameter's .grad is not a slice of the model's flat gradient buffer")
                if (g.data_ptr() - lo) % 16:
                    raise RuntimeError("FlatAdamW: a gradient slice of the flat buffer is not 16-byte aligned")
                offs.append((g.data_ptr() - lo) // 4)
            self._goff_host = offs
            self._t_goff = torch.tensor(offs, dtype=torch.int64, device=self._dev)
        else:   # cheap per-step check that the layout has not moved (first and last slice)
            lo = flat.data_ptr()
            gl = self._params[-1].grad
            if g0.data_ptr() - lo != 4 * self._goff_host[0] or gl is None or gl.data_ptr() - lo != 4 * self._goff_host[-1]:
                raise RuntimeError("FlatAdamW: the gradients are no longer slices of the flat gradient buffer in the recorded layout")
        return flat

    def validate(self) -> None:
        """Full check of what ``step`` only spot-checks (first / last tensor): every parameter still lives where the tables say and every
        ``.grad`` is the recorded slice of the model's flat gradient buffer.  ``CapturedTrainStep`` calls it once before capturing."""
        if [p.data_ptr() for p in self._params] != self._ptrs:
            raise RuntimeError("FlatAdamW: a parameter moved after the optimizer was built; build a new optimizer")
        flat = self._model.__dict__.get("_grad_flat")
        if flat is None or self._goff_host is None:
            return
        lo = flat.data_ptr()
        for p, off in zip(self._params, self._goff_host):
            if p.grad is None or p.grad.data_ptr() - lo != 4 * off:
                raise RuntimeError("FlatAdamW: a gradient is no longer the recorded slice of the flat gradient buffer")

    @torch.no_grad()
    def step(self, closure=None, max_norm: Optional[float] = None):
        """One AdamW update; ``max_norm``: clip the total gradient norm first (``clip_grad_norm_`` semantics, norm type 2).  Returns
        the total gradient norm before clipping as a device scalar (``None`` without ``max_norm``)."""
        if closure is not None:
            raise ValueError("FlatAdamW does not take a closure")
        if [p.data_ptr() for p in (self._params[0], self._params[-1])] != [self._ptrs[0], self._ptrs[-1]]:
            raise RuntimeError("FlatAdamW: the parameters moved (model.to(...) after the optimizer was built); build a new optimizer")
        flat = self._grad_base()
        g = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.refresh()
        t = L.AdamWTables(params=self._t_params.data_ptr(), grad_off=self._t_goff.data_ptr(), state_off=self._t_soff.data_ptr(),
                          numel=self._t_numel.data_ptr(), blocks=self._t_blocks.data_ptr(), ntensors=len(self._params),
                          nblocks=self._nblocks)
        with torch.cuda.device(self._dev):
            st = torch.cuda.current_stream(self._dev).cuda_stream
            L.check(self._lib.sepr_adamw_step(C.byref(t), flat.data_ptr(), flat.numel(), self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(),
                                              self._step.data_ptr(), self._lr.data_ptr(), float(g["betas"][0]), float(g["betas"][1]),
                                              float(g["eps"]), float(g["weight_decay"]), float(max_norm) if max_norm else 0.0,
                                              self._scal.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st), "sepr_adamw_step")
        return self._scal[0] if max_norm else None

    @property
    def clip_coef(self) -> torch.Tensor:
        """Device scalar: the clip coefficient of the last ``step(max_norm=...)``."""
        return self._scal[1]
