"""``bench.py --mode train``: one training step of the reference loop (engine.py:50-83) per "step" on the HIP training path.

step = model.train() forward (main + 4 auxiliary heads, batch-statistics BatchNorm, dropout at the model's configured rate)
       + PIT_SISNR_time on the main outputs + PIT_SISNR_mag on each auxiliary output + the 0.6 / 0.4 mix (engine.py:66-74)
       + backward through every block (sepr_*_bwd) + RCCL all-reduce of the flat gradient buffer when N > 1
       + clip_grad_norm_(5) + AdamW step (stock torch optimizer, as in the reference).
Inputs: synthetic two-speaker mixtures with their sources as targets, resident in HBM.  ``value`` = utterances/s over all ranks.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import torch

FP32_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS = 157.3, 2500.0
HBM_PEAK_GBS = 8000.0
TRAFFIC_SOURCE = "static: profiles/pmc_gemm_tn.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the contraction launches of one eager step, tools/pmc_traffic.sh); not re-measured in this run"


def _static_traffic(prec, launches):
    """HBM bytes per launch from the committed PMC collection (average over the 299 contractions of a batch-16 step)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_gemm_tn.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return rec.get(prec, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None
GFLOP_FWD = {"SepReformer_Base_WSJ0": 182.16, "SepReformer_Large_DM_WHAMR": 684.14}     # per 4 s utterance, forward incl. aux heads


XGMI_LINK_GBS = 153.0       # per-direction bandwidth of one xGMI link (7 per GPU, fully connected; MI355X_MICROARCH.md / SURVEY.md section 5)


def dp8_prediction(step_ms, batch, ar_bytes, ar_ms_here, world_here, overlapped):
    """What the first real 8-GPU run of this step should show, from pieces measured here (no multi-GPU node was available to the builder):
    per-rank compute = this run's step time (weak scaling: the per-rank batch does not change), plus the gradient all-reduce -
    ring model 2 (N-1)/N bytes / link bandwidth (RCCL's ring is per-link bound on point-to-point xGMI), lower bound = a direct
    reduce-scatter + all-gather over all 7 links - which in the captured step runs BETWEEN the two graph replays, i.e. not overlapped
    with the backward (the eager step overlaps the decoder half's bucket)."""
    n = 8
    ring_ms = 2.0 * (n - 1) / n * ar_bytes / (XGMI_LINK_GBS * 1e9) * 1e3
    direct_ms = 2.0 * (ar_bytes / n) / (XGMI_LINK_GBS * 1e9) * 1e3
    floor_ms = ar_ms_here if (ar_ms_here is not None and world_here == 1) else 0.0     # launch / RCCL-kernel floor measured on the 1-rank group
    exposed = max(ring_ms, floor_ms) * (0.5 if overlapped else 1.0)
    pred = step_ms + exposed - (floor_ms if world_here == 1 else 0.0)                   # this run's step already contains the 1-rank floor
    return {"n_gpus": n, "per_rank_step_ms_measured": round(step_ms, 3), "allreduce_bytes": int(ar_bytes),
            "allreduce_ms_measured_here": None if ar_ms_here is None else round(ar_ms_here, 3), "allreduce_ranks_here": world_here,
            "allreduce_ms_ring_model": round(ring_ms, 3), "allreduce_ms_direct_lower_bound": round(direct_ms, 3),
            "overlap": "decoder-half bucket under the encoder half of the backward (eager step)" if overlapped else
                       "none: the all-reduce runs between the two hipGraph replays of the captured step",
            "predicted_step_ms": round(pred, 3), "predicted_utt_per_s": round(n * batch / pred * 1e3, 1),
            "predicted_scaling_efficiency": round(step_ms / pred, 4),
            "note": "prediction, not a measurement: per-rank BatchNorm statistics, no parameter broadcast, one 58.8 MB all-reduce per step; host-side "
                    "launch jitter across 8 processes is not modelled"}


DTYPES = {
    "bf16x3": ("bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate); fp32 master weights, "
               "gradients and optimizer state"),
    "bf16": ("bf16 (plain bf16 operands in every projection, the fused GCFN forward, input-gradient projection and weight-gradient contraction: "
             "ONE bf16 MFMA per product, fp32 accumulate); fp32 master weights, gradients and optimizer state"),
    "fp32": "f32",
}


def run(variant, precision, B, steps, warmup, rank, world, dev, share, graphs=True):
    """One training measurement on this rank's device; the process group (if any) is already initialised.  Returns the record
    (rank 0) or None."""
    from . import dist as sdist
    from . import lib as L
    from .config import VARIANTS
    from .criterion import PIT_SISNR_mag, PIT_SISNR_time
    from .model import Model
    from .synth import synth_sources

    lib = L.load()
    cfg = VARIANTS[variant]
    model = Model.from_config(cfg, init_seed=0, precision=precision).load_synthetic_(0).to(dev)
    model.train()
    # launch mode of the step: "step" = the WHOLE step as hipGraph replays (train_step.CapturedTrainStep: forward + criteria + backward;
    # RCCL all-reduce; clip + optimizer), "split" = the separator's forward / backward as graphs with eager criteria / clip / optimizer
    # (model._TrainGraph), "off" = every kernel launched from the host.  True selects "step" with "split" as the fallback.
    mode = {True: "step", False: "off", None: "off"}.get(graphs, graphs)
    model.train_graphs = mode == "split"
    sync = sdist.GradSync()
    model.grad_sync = sync
    samples = 32000
    src = torch.from_numpy(synth_sources(B, samples, seed=4321 + rank * B)).to(dev)
    x = src.sum(1).contiguous()
    targets = [src[:, s].contiguous() for s in range(cfg.num_spks)]
    sizes = torch.full((B,), samples)
    crit_t = PIT_SISNR_time(dev, cfg.num_spks, True)
    crit_m = PIT_SISNR_mag(dev, 512, 128, "hann", cfg.num_stages, cfg.num_spks, True, False)
    params = list(model.parameters())
    # the reference's AdamW + clip_grad_norm_ (engine.py:76-77): sepreformer_amd.optim.FlatAdamW runs both as three launches over the
    # flat gradient buffer (same arithmetic; tests/test_train_gpu.py compares it with torch.optim.AdamW); SEPR_BENCH_OPT=torch: torch's
    flat_opt = os.environ.get("SEPR_BENCH_OPT", "flat") != "torch"
    if flat_opt:
        from .optim import FlatAdamW
        opt = FlatAdamW(model, lr=1.0e-4, weight_decay=1.0e-2)
    else:
        try:
            opt = torch.optim.AdamW(params, lr=1.0e-4, weight_decay=1.0e-2, fused=True, capturable=(mode == "step"))
        except (TypeError, RuntimeError):
            opt = torch.optim.AdamW(params, lr=1.0e-4, weight_decay=1.0e-2, capturable=(mode == "step"))
    last = {}

    def loss_fn(audio, aux, *tg):
        tg = list(tg)
        l_time = crit_t(estims=audio, input_sizes=sizes, target_attr=tg)
        l_mag = [crit_m(estims=a, idx=i, input_sizes=sizes, target_attr=tg) for i, a in enumerate(aux)]
        return ((1 - 0.4) * l_time + 0.4 * sum(l_mag) / len(l_mag)) / cfg.num_spks             # engine.py:72-74

    def host_step():
        opt.zero_grad(set_to_none=True)
        audio, aux = model(x)
        loss = loss_fn(audio, aux, *targets)
        loss.backward()
        if flat_opt:
            gn = opt.step(max_norm=5.0)                                                       # engine.py:76-77 in one call
        else:
            gn = torch.nn.utils.clip_grad_norm_(params, 5.0)                                  # engine.py:76
            opt.step()
        last["loss"], last["gn"] = loss.detach(), gn

    captured, capture_error = None, None
    if mode == "step":
        try:
            from .train_step import CapturedTrainStep
            captured = CapturedTrainStep(model, loss_fn, opt, x, targets, max_norm=5.0, warmup=max(warmup, 1))
        except Exception as e:                      # noqa: BLE001 - a box that cannot capture the whole step still gets measured
            capture_error = f"{type(e).__name__}: {e}"[:300]
            mode = "split"
            model.train_graphs = True

    def step():
        if captured is not None:
            last["loss"], last["gn"] = captured(x, targets)
        else:
            host_step()

    if captured is None:
        for _ in range(max(warmup, 1) if mode == "split" else warmup):     # (split mode: the first step captures)
            step()
    torch.cuda.synchronize(dev)
    graphs = mode != "off"
    if not graphs:
        L.check(lib.sepr_prof_start(L.SITE_WGRAD, 400 * (max(steps, 1) + 1) + 8), "sepr_prof_start")
    sdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0            # host-side enqueue time of the K steps (no synchronisation inside a step)
    torch.cuda.synchronize(dev)
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    # host cost of a step on its own: one more step enqueued to an IDLE device, outside the timed region (inside it the host runs
    # ahead of the device and the wall time of its loop includes waiting for queue slots - 16 ms at 2 steps, 54 ms at 4)
    te0 = time.perf_counter()
    step()
    t_enq = time.perf_counter() - te0
    torch.cuda.synchronize(dev)
    n_l, ms, fl = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
    if graphs:
        # a replayed graph has no per-launch events: the dominant kernel's launch durations come from ONE extra eager step
        # of the same model outside the timed region (same kernels, same shapes)
        model.train_graphs = False
        L.check(lib.sepr_prof_start(L.SITE_WGRAD, 400 + 8), "sepr_prof_start")
        host_step()
        torch.cuda.synchronize(dev)
        model.train_graphs = mode == "split"
    L.check(lib.sepr_prof_stop(C.byref(n_l), C.byref(ms), C.byref(fl)), "sepr_prof_stop")
    algo_bytes = float(lib.sepr_prof_last_bytes())
    # the LARGEST kernel of the step - the GCFN backward's middle kernel - timed the same way on one more eager step (its own site)
    n_m, ms_m, fl_m, bytes_m = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0), 0.0
    if getattr(model, "fused_gcfn_train", True):
        prev_graphs = model.train_graphs
        model.train_graphs = False
        L.check(lib.sepr_prof_start(L.SITE_GCFN_BWD, 128), "sepr_prof_start")
        host_step()
        torch.cuda.synchronize(dev)
        model.train_graphs = prev_graphs
        L.check(lib.sepr_prof_stop(C.byref(n_m), C.byref(ms_m), C.byref(fl_m)), "sepr_prof_stop")
        bytes_m = float(lib.sepr_prof_last_bytes())
    rank_min_s, rank_max_s = sdist.min_max_over_ranks(elapsed, dev)     # per-rank spread before the max: launch jitter between ranks
    elapsed = sdist.max_over_ranks(elapsed, dev)
    # the gradient all-reduce on its own (the collective of this path): wall time of one synchronised call on the flat buffer,
    # median of 5 - at world size 1 the floor RCCL adds to a step, at N > 1 the real exchange
    ar_ms = None
    if sync.calls:
        probe = torch.zeros(sum(p_.numel() for p_ in params), dtype=torch.float32, device=dev)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            sdist.all_reduce_off_stream(probe)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - ta)
        ar_ms = 1e3 * sorted(ts[1:])[2]
        del probe
    rec = None
    if rank == 0:
        utt_per_s = world * B * steps / elapsed
        prec = model.precision
        peak = FP32_MFMA_PEAK_TFLOPS if prec == "fp32" else BF16_MFMA_PEAK_TFLOPS
        mult = 3.0 if prec == "bf16x3" else 1.0
        sec = ms.value / 1e3
        algo_tf = fl.value / 1e12 / sec if sec > 0 else 0.0
        gflop = 3.0 * GFLOP_FWD.get(variant, 0.0)                      # forward + input gradients + weight gradients
        rec = {
            "metric": f"training utterances/sec (4 s, 8 kHz, 2-spk) {variant}: forward + PIT SI-SNR losses + backward + clip + AdamW",
            "value": round(utt_per_s, 3), "unit": "utt/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / max(steps, 1), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[prec], "data": "synthetic",
            "config": {"workload": f"{variant} training step, batch={B} per GPU, 4 s @ 8 kHz, 2 speakers (BASELINE.json configs[4])",
                       "batch_per_gpu": B, "samples": samples, "precision": prec, "dropout": model.dropout_p,
                       "dropout_sites": "all the reference's sites: GCFN x2, CLA, attention probabilities + attention output (EGA and speaker attention)",
                       "step_launch": {"step": "two hipGraph replays per step: weight re-pack + forward + criteria + backward | RCCL all-reduce (eager) | "
                                               "clip + optimizer (train_step.CapturedTrainStep)",
                                       "split": "two hipGraph replays per step (weight re-pack + forward; backward) + eager criteria / clip / optimizer",
                                       "off": "eager (every kernel launched from the host)"}[mode],
                       "optimizer": type(opt).__name__ + (" (fused)" if getattr(opt, "defaults", {}).get("fused") else "") +
                                    (" = clip_grad_norm_ + AdamW as 3 launches over the flat gradient buffer" if flat_opt else ""),
                       "clip_norm": 5.0, "parallelism": f"data-parallel x{world}, flat-buffer RCCL all-reduce" + (" (DEBUG: all ranks share GPU 0, gloo collective)" if share else "")},
            "capture_fallback": capture_error,
            "host_enqueue_ms_per_step": round(1e3 * t_enq, 3),
            "host_loop_ms_per_step": round(1e3 * t_host / max(steps, 1), 3),
            "loss": round(float(last["loss"]), 4), "grad_norm": round(float(last["gn"]), 4),
            "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "collective_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            "allreduce_bytes_per_step": (sync.bytes // max(sync.calls, 1)) if sync.calls else 0,
            "per_rank_ms_per_step": {"min": round(1e3 * rank_min_s / max(steps, 1), 3), "max": round(1e3 * rank_max_s / max(steps, 1), 3), "ranks": world},
            "grad_allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
            "model_tflops": round(utt_per_s * gflop / 1e3 / world, 2),
            "model_frac_algorithmic": round(utt_per_s * gflop / 1e3 / world / peak, 4),
            "dp8_prediction": dp8_prediction(1e3 * elapsed / max(steps, 1), B, (sync.bytes // max(sync.calls, 1)) if sync.calls else 0, ar_ms, world,
                                             overlapped=(mode == "off")),
            # The contraction over M = batch x frames rows reads both operands once for a [N,K] result with N, K <= 1024: at
            # 2 N K / (4 (N + K)) = 50-130 FLOP per byte it sits under the ridge of the bf16 MFMA (312 FLOP/B) - HBM is its roofline.
            # achieved = algorithmic bytes of the timed launches / their hipEvent durations; the matrix-pipe view rides along.
            "roofline": {"kernel": "gemm_tn_kernel / gemm_tnd_kernel (weight-gradient contraction G[N][K] = sum_m dY[m][n] X[m][k], all 299 projections of a step; two bf16 operands: LDS-DMA ring + ds_read_b64_tr_b16)",
                         "bound": "hbm", "achieved": round(algo_bytes / 1e9 / sec, 1) if sec > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo_bytes / 1e9 / sec / HBM_PEAK_GBS, 4) if sec > 0 else 0.0,
                         "algorithmic_tflops": round(algo_tf, 2), "mfma_pipe_frac": round(mult * algo_tf / peak, 4),
                         "traffic": _static_traffic(prec, n_l.value), "traffic_source": TRAFFIC_SOURCE,
                         "algorithmic_bytes_per_launch": round(algo_bytes / max(n_l.value, 1)), "launches": int(n_l.value),
                         "avg_launch_ms": round(ms.value / max(n_l.value, 1), 4),
                         "measured_on": "one eager step after the timed region (a replayed hipGraph has no per-launch events)" if graphs else "the timed steps"},
        }
        if n_m.value > 0 and ms_m.value > 0:
            sec_m = ms_m.value / 1e3
            tf_m = fl_m.value / 1e12 / sec_m
            # recomputed up-projection + input gradient of net2.2 (+ the conv / GLU forward and backward on the VALU): matrix-pipe bound in
            # principle (18 F^2 FLOP per row against ~2.8 KB per row), so the fraction is quoted against the dense bf16 MFMA peak; the HBM view rides along
            rec["roofline_gcfn_bwd"] = {
                "kernel": "gcfn_bwd_mid_kernel (GCFN backward middle: recomputes LN(x) W1, dgd = dy (ls W2), conv / GLU / dropout forward + backward; the largest kernel of the step)",
                "bound": "mfma", "achieved": round(tf_m, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf_m / peak, 4),
                "mfma_pipe_frac": round(mult * tf_m / peak, 4), "ceiling": round(1.0 / mult, 4),
                "launches": int(n_m.value), "avg_launch_ms": round(ms_m.value / n_m.value, 4),
                "algorithmic_bytes_per_launch": round(bytes_m / n_m.value), "hbm_gbs": round(bytes_m / 1e9 / sec_m, 1),
                "hbm_frac": round(bytes_m / 1e9 / sec_m / HBM_PEAK_GBS, 4), "traffic": None,
                "measured_on": "one more eager step after the timed region (hipEvents on the launch stream)"}
    model.grad_sync = None
    del opt, model
    torch.cuda.empty_cache()
    return rec


def main(args):
    from . import dist as sdist

    torch.manual_seed(0)        # the dropout seeds derive from torch.initial_seed(): `loss` / `grad_norm` are box-independent fingerprints
    share = bool(getattr(args, "share_gpu", False))   # DEBUG: all ranks on GPU 0, gloo collectives (exercises the N > 1 code path on one GPU)
    # a process group also at world size 1: the gradient all-reduce then really runs through RCCL on a 1-GPU box
    rank, world, local = sdist.init_from_env("gloo" if share else None, single_rank_group=not share)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the separator path)")
    if share:
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    graphs = getattr(args, "train_graphs", None) or "step"
    if getattr(args, "no_train_graphs", False) or os.environ.get("SEPR_TRAIN_GRAPHS", "1") == "0":
        graphs = "off"
    rec = run(args.variant, args.precision, args.batch or 8, args.steps, args.warmup, rank, world, dev, share, graphs)
    if rank == 0:
        emit = getattr(args, "_emit", None)
        if emit is not None:
            emit(rec)
        else:
            print(json.dumps(rec), flush=True)
    sdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
