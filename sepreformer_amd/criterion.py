"""Device-side permutation-invariant SI-SNR criteria behind the reference's criterion call surface.

Mirrors ``utils/implements/criterions.py`` of the reference:

* ``PIT_SISNR_time(device, num_spks, scale_inv)(estims=..., input_sizes=..., target_attr=...)`` -> scalar loss
  (reference :180-217, called by ``engine.py:70,103``);
* ``PIT_SISNRi(device, num_spks, scale_inv)(estims=..., mixture=..., input_sizes=..., target_attr=..., eps=...)``
  -> ``(mean summed improvement, per-speaker improvements)`` (reference :220-260, called by ``engine.py:131``).

``estims`` / ``target_attr`` are lists of ``[B,T]`` tensors (or one ``[S,B,T]`` tensor) on the HIP device.  The
arithmetic is one pass over the waveforms in ``csrc/sepr_criterion.hip`` through ``sepr_pit_sisnr_fwd``; there is no
CPU implementation here (the CPU restatement is ``oracle/criterion_oracle.py``, test infrastructure).

Training: when an estimate requires grad, ``PIT_SISNR_time`` / ``PIT_SISNR_mag`` return an
autograd-connected scalar; their backward is ``sepr_pit_sisnr_bwd`` / ``sepr_pit_sisnr_mag_bwd`` (closed form from the same
moments; the STFT adjoint is one more projection with the transposed DFT kernel), the permutation chosen in the forward
is held fixed, as ``torch.min`` does in the reference.
"""
from __future__ import annotations

from typing import List, Sequence, Union

import torch

from . import lib as L

_TensorList = Union[torch.Tensor, Sequence[torch.Tensor]]


def _needs_grad(x: _TensorList) -> bool:
    if not torch.is_grad_enabled():
        return False
    return x.requires_grad if isinstance(x, torch.Tensor) else any(t.requires_grad for t in x)


def _stack_g(x: _TensorList) -> torch.Tensor:
    """Stack keeping the autograd graph (the training criteria differentiate through this)."""
    t = x if isinstance(x, torch.Tensor) else torch.stack(list(x), dim=0)
    return t.to(torch.float32).contiguous()


class _PitTimeFn(torch.autograd.Function):
    """sum_b w_b * loss_b of PIT_SISNR_time with d/d est from sepr_pit_sisnr_bwd (criterions.py:191-217)."""

    @staticmethod
    def forward(ctx, est, tgt, eps, clamp_min):
        out = pit_sisnr(est, tgt, eps_loss=eps, clamp_min=clamp_min)
        ctx.save_for_backward(est, tgt, out["loss_perm"])
        ctx.consts = (eps, clamp_min)
        return out["loss"]

    @staticmethod
    def backward(ctx, gl):
        est, tgt, perm = ctx.saved_tensors
        eps, clamp_min = ctx.consts
        S, B, T = est.shape
        lib = L.load()
        with torch.cuda.device(est.device):
            nbytes = lib.sepr_workspace_bytes(L.OP_PIT, B, T, 0, 0, 0, S) + 16 * B * S + 512
            ws = torch.empty(nbytes, dtype=torch.uint8, device=est.device)
            dest = torch.empty_like(est)
            glc = gl.detach().to(torch.float32).contiguous()
            L.check(lib.sepr_pit_sisnr_bwd(est.data_ptr(), tgt.data_ptr(), perm.data_ptr(), glc.data_ptr(), S, B, T, eps, clamp_min,
                                           dest.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(est.device).cuda_stream),
                    "sepr_pit_sisnr_bwd")
        return dest, None, None, None


class _PitMagFn(torch.autograd.Function):
    """Per-utterance PIT_SISNR_mag loss with d/d est from sepr_pit_sisnr_mag_bwd (criterions.py:148-176)."""

    @staticmethod
    def forward(ctx, est, tgt, dft, dft_t, frame_len, frame_shift, eps):
        out = pit_sisnr_mag(est, tgt, dft, frame_len, frame_shift, eps)
        ctx.save_for_backward(est, tgt, out["perm"], dft, dft_t)
        ctx.consts = (frame_len, frame_shift, eps)
        return out["loss"]

    @staticmethod
    def backward(ctx, gl):
        est, tgt, perm, dft, dft_t = ctx.saved_tensors
        frame_len, frame_shift, eps = ctx.consts
        S, B, T = est.shape
        lib = L.load()
        with torch.cuda.device(est.device):
            nbytes = lib.sepr_pit_sisnr_mag_bwd_workspace(S, B, T, frame_len, frame_shift)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=est.device)
            dest = torch.empty_like(est)
            glc = gl.detach().to(torch.float32).contiguous()
            L.check(lib.sepr_pit_sisnr_mag_bwd(est.data_ptr(), tgt.data_ptr(), perm.data_ptr(), glc.data_ptr(), S, B, T, dft.data_ptr(),
                                               dft_t.data_ptr(), frame_len, frame_shift, eps, dest.data_ptr(), ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream(est.device).cuda_stream), "sepr_pit_sisnr_mag_bwd")
        return dest, None, None, None, None, None, None


def _stack(x: _TensorList, what: str) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.stack(list(x), dim=0)
    if t.dim() != 3:
        raise RuntimeError(f"{what}: expected num_spks tensors of shape [batch, samples]")
    if not t.is_cuda:
        raise RuntimeError(f"{what} is not on the HIP device (no CPU fallback exists)")
    return t.detach().to(torch.float32).contiguous()


def pit_sisnr(estims: _TensorList, targets: _TensorList, mixture: torch.Tensor = None, eps_loss: float = 1.0e-8,
              eps_i: float = 1.0e-15, clamp_min: float = -30.0):
    """One launch pair over ``[S,B,T]`` estimates / targets.  Returns a dict with ``loss`` ``[B]`` (PIT_SISNR_time per
    utterance), ``loss_perm`` ``[B,S]`` and, when ``mixture`` is given, ``sisnri`` ``[B,S]`` / ``sisnri_perm`` ``[B,S]``."""
    est, tgt = _stack(estims, "estims"), _stack(targets, "target_attr")
    if est.shape != tgt.shape:
        raise RuntimeError(f"estims {tuple(est.shape)} and targets {tuple(tgt.shape)} differ")
    S, B, T = est.shape
    dev = est.device
    mix = None
    if mixture is not None:
        mix = mixture.detach().to(torch.float32).contiguous()
        if tuple(mix.shape) != (B, T) or mix.device != dev:
            raise RuntimeError("mixture must be [batch, samples] on the same device as the estimates")
    lib = L.load()
    with torch.cuda.device(dev):
        nbytes = lib.sepr_workspace_bytes(L.OP_PIT, B, T, 0, 0, 0, S)
        if nbytes == 0:
            raise RuntimeError(f"unsupported PIT problem: num_spks={S}, batch={B}, samples={T}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        loss_perm = torch.empty(B, S, dtype=torch.int32, device=dev)
        sisnri = torch.empty(B, S, dtype=torch.float32, device=dev) if mix is not None else None
        sisnri_perm = torch.empty(B, S, dtype=torch.int32, device=dev) if mix is not None else None
        L.check(lib.sepr_pit_sisnr_fwd(
            est.data_ptr(), tgt.data_ptr(), None if mix is None else mix.data_ptr(), S, B, T, eps_loss, eps_i, clamp_min,
            loss.data_ptr(), loss_perm.data_ptr(), None if sisnri is None else sisnri.data_ptr(),
            None if sisnri_perm is None else sisnri_perm.data_ptr(), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(dev).cuda_stream), "sepr_pit_sisnr_fwd")
    return {"loss": loss, "loss_perm": loss_perm, "sisnri": sisnri, "sisnri_perm": sisnri_perm}


def stft_kernel(frame_len: int, frame_hop: int, window: str = "hann") -> torch.Tensor:
    """The reference's conv-STFT kernel (``STFTBase._init_kernel``, criterions.py:43-61) as a ``[frame_len + 2 (padded to
    a multiple of 4), frame_len]`` matrix: windowed, scaled real DFT rows, then imaginary rows, then zero rows.  A
    constant of the criterion (built on the host once, like the reference does in ``__post_init__``)."""
    if window != "hann":
        raise NotImplementedError("only the 'hann' window is configured by the reference")
    N = frame_len
    W = torch.hann_window(frame_len)
    if N // 4 == frame_hop:
        W = (2 / 3) ** 0.5 * W
    elif N // 2 == frame_hop:
        W = W ** 0.5
    S = 0.5 * (N * N / frame_hop) ** 0.5
    K = torch.fft.rfft(torch.eye(N) / S, dim=1)[:frame_len]                 # [N, N/2+1]
    K = torch.stack((torch.real(K), torch.imag(K)), dim=2)                  # [N, N/2+1, 2]
    K = torch.transpose(K, 0, 2) * W                                         # [2, N/2+1, N]
    K = torch.reshape(K, (N + 2, frame_len))
    pad = (-(N + 2)) % 4
    return torch.cat([K, torch.zeros(pad, frame_len)], 0).to(torch.float32).contiguous()


def pit_sisnr_mag(estims: _TensorList, targets: _TensorList, dft: torch.Tensor, frame_len: int, frame_shift: int,
                  eps: float = 1.0e-12):
    """Per-utterance PIT_SISNR_mag loss ``[B]`` and its permutation ``[B,S]`` (one moment pass, one f32-MFMA STFT
    projection over all 2S waveforms, one pair-sum pass)."""
    est, tgt = _stack(estims, "estims"), _stack(targets, "target_attr")
    if est.shape != tgt.shape:
        raise RuntimeError(f"estims {tuple(est.shape)} and targets {tuple(tgt.shape)} differ")
    S, B, T = est.shape
    dev = est.device
    lib = L.load()
    with torch.cuda.device(dev):
        nbytes = lib.sepr_pit_sisnr_mag_workspace(S, B, T, frame_len, frame_shift)
        if nbytes == 0:
            raise RuntimeError(f"unsupported PIT_SISNR_mag problem: num_spks={S}, batch={B}, samples={T}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        perm = torch.empty(B, S, dtype=torch.int32, device=dev)
        L.check(lib.sepr_pit_sisnr_mag_fwd(est.data_ptr(), tgt.data_ptr(), S, B, T, dft.data_ptr(), frame_len, frame_shift, eps,
                                           loss.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream(dev).cuda_stream), "sepr_pit_sisnr_mag_fwd")
    return {"loss": loss, "perm": perm}


class _Base:
    def __init__(self, device, num_spks: int, scale_inv: bool = True):
        if not scale_inv:
            raise NotImplementedError("only the scale-invariant form is built (scale_inv: true in every shipped config)")
        self.device, self.num_spks, self.scale_inv = torch.device(device), num_spks, scale_inv

    def __repr__(self):
        return f"<{type(self).__name__}(device={self.device!r}, num_spks={self.num_spks!r}, scale_inv={self.scale_inv!r})>"

    def _check(self, estims):
        n = estims.shape[0] if isinstance(estims, torch.Tensor) else len(estims)
        if n != self.num_spks:
            raise RuntimeError(f"expected {self.num_spks} estimates, got {n}")


class PIT_SISNR_time(_Base):
    def __call__(self, **kwargs) -> torch.Tensor:
        estims, targets = kwargs["estims"], kwargs["target_attr"]
        self._check(estims)
        if _needs_grad(estims):                                                   # training: autograd-connected (engine.py:70,75)
            loss = _PitTimeFn.apply(_stack_g(estims), _stack(targets, "target_attr"), 1.0e-8, -30.0)
            return torch.sum(loss) / kwargs["input_sizes"].shape[0]
        out = pit_sisnr(estims, targets)
        return torch.sum(out["loss"]) / kwargs["input_sizes"].shape[0]           # reference :216-217


class PIT_SISNRi(_Base):
    def __call__(self, **kwargs):
        estims, targets = kwargs["estims"], kwargs["target_attr"]
        self._check(estims)
        out = pit_sisnr(estims, targets, mixture=kwargs["mixture"], eps_i=kwargs["eps"])
        per = out["sisnri"]
        mean = torch.sum(per) / kwargs["input_sizes"].shape[0]                    # reference :255-257
        return mean, (per[0] if per.shape[0] == 1 else per)


class PIT_SISNR_mag:
    """``PIT_SISNR_mag(device, frame_length, frame_shift, window, num_stages, num_spks, scale_inv, mel_opt)`` of the
    reference (criterions.py:117-176): ``__call__(estims=..., idx=..., input_sizes=..., target_attr=...)`` -> scalar loss.
    ``idx`` selects one of ``num_stages`` identical STFT layers in the reference; there is one kernel matrix here."""

    def __init__(self, device, frame_length: int, frame_shift: int, window: str, num_stages: int, num_spks: int,
                 scale_inv: bool = True, mel_opt: bool = False):
        if not scale_inv or mel_opt:
            raise NotImplementedError("only scale_inv=True, mel_opt=False (what every shipped config uses) is built")
        self.device = torch.device(device)
        self.frame_length, self.frame_shift, self.window = frame_length, frame_shift, window
        self.num_stages, self.num_spks, self.scale_inv, self.mel_opt = num_stages, num_spks, scale_inv, mel_opt
        self._dft = stft_kernel(frame_length, frame_shift, window)
        ldd = (frame_length + 2 + 31) // 32 * 32                                   # K of the adjoint projection (include/sepr.h)
        self._dft_t = torch.zeros(frame_length, ldd)
        self._dft_t[:, : frame_length + 2] = self._dft[: frame_length + 2].t()
        if self.device.type == "cuda":
            self._dft = self._dft.to(self.device)
            self._dft_t = self._dft_t.to(self.device).contiguous()

    def __repr__(self):
        return (f"<PIT_SISNR_mag(device={self.device!r}, frame_length={self.frame_length}, frame_shift={self.frame_shift}, "
                f"window={self.window!r}, num_stages={self.num_stages}, num_spks={self.num_spks}, scale_inv=True, mel_opt=False)>")

    def __call__(self, **kwargs) -> torch.Tensor:
        estims, targets = kwargs["estims"], kwargs["target_attr"]
        if not 0 <= int(kwargs["idx"]) < self.num_stages:
            raise IndexError("idx out of range")                                    # self.stft[idx] in the reference
        n = estims.shape[0] if isinstance(estims, torch.Tensor) else len(estims)
        if n != self.num_spks:
            raise RuntimeError(f"expected {self.num_spks} estimates, got {n}")
        if self._dft.device.type != "cuda":
            raise RuntimeError("PIT_SISNR_mag was built for a non-HIP device (no CPU fallback exists)")
        if _needs_grad(estims):                                                     # training (engine.py:68,75)
            loss = _PitMagFn.apply(_stack_g(estims), _stack(targets, "target_attr"), self._dft, self._dft_t, self.frame_length,
                                   self.frame_shift, 1.0e-12)
            return torch.sum(loss) / kwargs["input_sizes"].shape[0]
        out = pit_sisnr_mag(estims, targets, self._dft, self.frame_length, self.frame_shift)
        return torch.sum(out["loss"]) / kwargs["input_sizes"].shape[0]             # reference :175-176
