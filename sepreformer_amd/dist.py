"""Multi-GPU inference: utterances are independent in eval mode, so the batch is sharded by utterance.

One process per GPU (``torch.distributed``; backend ``nccl`` == RCCL over xGMI on ROCm, ``gloo`` for the
CPU tests).  The data path has NO collective: each rank separates its own contiguous slice with weights
replicated once at load.  (The reference instead uses single-process ``torch.nn.parallel.data_parallel``,
``engine.py:64,98,130,167``, which re-broadcasts all 14.7 M parameters on every forward.)  The only
optional exchanges are an all-reduce of three metric scalars, or an all-gather of the separated
waveforms when one rank must hold them all.
"""
from __future__ import annotations

import datetime
import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice ``[start, stop)`` of ``total`` utterances for ``rank``."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  With WORLD_SIZE absent or 1 no process group is created unless
    ``single_rank_group`` is set: then a ONE-rank group is initialised (127.0.0.1, a free port), so that the path's
    collectives (metric all-reduce, gradient all-reduce) really go through RCCL on a 1-GPU box as well."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # bounded collective timeout: a dead rank must fail the job, not hang it (torch's defaults are 600 s nccl / 1800 s gloo);
        # the longest legitimate wait between collectives here is a whole-step hipGraph capture (~30 s).  SEPR_DIST_TIMEOUT_S overrides.
        kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("SEPR_DIST_TIMEOUT_S", "300") or 300))}
        if world == 1:
            global _SINGLE_RANK_COLLECTIVES
            _SINGLE_RANK_COLLECTIVES = True      # the caller asked for a one-rank group in order to exercise the collectives
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


# A ONE-rank process group runs the path's collectives only when it was created for that purpose by
# ``init_from_env(single_rank_group=True)`` (bench.py: RCCL is then exercised on a 1-GPU box); a one-rank group made by anything else
# (a launcher with WORLD_SIZE=1) skips them - they would be pure overhead.
_SINGLE_RANK_COLLECTIVES = False


def collectives_active(group=None) -> bool:
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _SINGLE_RANK_COLLECTIVES)


def all_reduce_off_stream(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> None:
    """``dist.all_reduce`` whose completion event lives on the backend's INTERNAL stream, with the caller's stream ordered behind it.

    torch runs a synchronous collective (``async_op=False``) on the caller's CURRENT stream and records the work's completion event there;
    the process group's watchdog thread then polls that event (``hipEventQuery``, every 100 ms) until it has retired the work.  If the same
    stream starts - or is pulled into - a hipGraph capture before that, HIP refuses the query ("operation not permitted on an event last
    recorded in a capturing stream", even for an event recorded BEFORE the capture began), the watchdog thread throws and the process
    aborts: the ~1-in-15 SIGABRT of round 4's training sub-runs (backtrace: profiles/r05_fault_rccl_watchdog_backtrace.txt; DESIGN.md
    section 10c).  ``async_op=True`` + ``wait()`` keeps the event on the backend's own stream, which never captures; ``wait()`` is a
    stream-level dependency, not a host block.  gloo (CPU tests) completes inside ``wait()`` as well."""
    work = dist.all_reduce(t, op=op, group=group, async_op=True)
    if work is not None:
        work.wait()


def reduce_metric_sums(values: torch.Tensor) -> torch.Tensor:
    """Sum a small vector of metric accumulators (e.g. [sum SI-SNR, sum SI-SNRi, count]) over ranks."""
    if collectives_active():
        all_reduce_off_stream(values)
    return values


def max_over_ranks(seconds: float, device: torch.device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        all_reduce_off_stream(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_max_over_ranks(seconds: float, device: torch.device) -> Tuple[float, float]:
    """(min, max) of a per-rank wall time: the spread IS the host-side launch jitter between ranks (the one thing a model of the
    collective cannot predict); one 2-element MAX all-reduce of (-t, t)."""
    t = torch.tensor([-seconds, seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        all_reduce_off_stream(t, op=dist.ReduceOp.MAX)
    lo, hi = t.tolist()
    return -lo, hi


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":          # (RCCL's barrier is a synchronous all-reduce on the caller's stream: see all_reduce_off_stream)
            t = torch.zeros(1, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
            all_reduce_off_stream(t)
            torch.cuda.current_stream().synchronize()
        else:
            dist.barrier()


def separate_sharded(separate: Callable[[torch.Tensor], torch.Tensor], mixtures: torch.Tensor,
                     gather: bool = False, chunk: int = 32) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Run ``separate`` (``[b,T] -> [S,b,T']``) on this rank's slice of ``mixtures`` ``[B,T]``.

    Returns ``(out, (start, stop))``: ``out`` is this rank's ``[S, stop-start, T']``, or with
    ``gather=True`` the full ``[S,B,T']`` on every rank (all-gather of equal-padded shards).
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = mixtures.shape[0]
    start, stop = shard_range(B, rank, world)
    outs: List[torch.Tensor] = []
    for s in range(start, stop, chunk):
        outs.append(separate(mixtures[s:min(stop, s + chunk)]))
    local = torch.cat(outs, dim=1) if outs else None
    if not gather or world == 1:
        if local is None:
            raise RuntimeError("rank has no utterances and gather=False")
        return local, (start, stop)
    # all ranks must agree on the padded shard size; ranks with an empty slice learn S/T' from rank 0
    per = (B + world - 1) // world
    meta = torch.zeros(2, dtype=torch.long, device=mixtures.device)
    if local is not None:
        meta[0], meta[1] = local.shape[0], local.shape[2]
    all_reduce_off_stream(meta, op=dist.ReduceOp.MAX)
    S, Tout = int(meta[0]), int(meta[1])
    pad = torch.zeros(S, per, Tout, dtype=torch.float32, device=mixtures.device)
    if local is not None:
        pad[:, : stop - start] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    full = torch.cat([parts[r][:, : shard_range(B, r, world)[1] - shard_range(B, r, world)[0]] for r in range(world)], dim=1)
    return full, (start, stop)


class GradSync:
    """Data-parallel gradient averaging for the training path (BASELINE configs[4]: DP over the GPUs of one node).

    The reference trains with single-process ``torch.nn.parallel.data_parallel`` (engine.py:64): every step it
    re-broadcasts all 14.7 M parameters and reduce-adds the replicas' gradients onto GPU 0.  Here every rank owns a full
    replica and its slice of the batch; the training backward leaves ALL parameter gradients in one flat fp32 buffer
    (``train_pack.GradBuffer``, 58.8 MB for Base) laid out in parameter order, so the exchange is TWO large RCCL all-reduces
    over xGMI instead of 710 small ones, and no per-step parameter broadcast:

    * ``begin(flat, offset)`` - called by the backward as soon as the decoder half of the U-Net is done (its parameters,
      ~60 % of the model, are the tail ``flat[offset:]`` of the buffer): an asynchronous all-reduce that runs on RCCL's
      stream underneath the encoder half of the backward;
    * ``__call__(flat)`` - at the end of the backward: all-reduce of the head ``flat[:offset]`` (encoder, shared tables),
      wait for the first one, then the 1/world scale that turns the sum of per-rank batch-mean losses into the global
      batch mean.

    xGMI is point-to-point (7 links per GPU), so a ring all-reduce is per-link bound: two ~25-35 MB messages keep every
    link busy with large transfers, whereas per-tensor reductions would be latency-bound.  BatchNorm statistics stay per
    rank, like the reference's per-replica statistics.  Install with ``model.grad_sync = GradSync()``.
    """

    def __init__(self, group=None, overlap: bool = True):
        self.group, self.overlap = group, overlap
        self.calls = 0
        self.bytes = 0
        self._pending = None

    def _active(self) -> bool:
        return collectives_active(self.group)      # one rank: only in a group made by init_from_env(single_rank_group=True)

    def abort(self) -> None:
        """Drop an early all-reduce whose backward did not finish (exception between ``begin`` and ``__call__``)."""
        if self._pending is not None:
            work, _ = self._pending
            self._pending = None
            try:
                work.wait()
            except Exception:      # noqa: BLE001 - the step is already failing; the handle must not leak into the next one
                pass

    def begin(self, flat: torch.Tensor, offset: int) -> None:
        self.abort()               # a stale handle of a failed step must never be consumed by this one
        if not (self._active() and self.overlap) or offset <= 0 or offset >= flat.numel():
            return
        tail = flat[offset:]
        self._pending = (dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group, async_op=True), offset)
        self.bytes += tail.numel() * tail.element_size()

    def __call__(self, flat: torch.Tensor) -> None:
        if not self._active():
            return
        if self._pending is not None:
            work, offset = self._pending
            self._pending = None
            head = flat[:offset]
            all_reduce_off_stream(head, group=self.group)
            work.wait()
            self.bytes += head.numel() * head.element_size()
        else:
            all_reduce_off_stream(flat, group=self.group)          # (never on the caller's stream: see all_reduce_off_stream)
            self.bytes += flat.numel() * flat.element_size()
        if dist.get_world_size(self.group) > 1:
            flat.div_(dist.get_world_size(self.group))
        self.calls += 1
