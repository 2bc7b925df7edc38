#!/usr/bin/env python3
"""Throughput of the SepReformer-Base separator forward on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full ``Model.forward`` (main + the 4 auxiliary heads, exactly what the reference's
forward evaluates, model.py:38-54) over one batch of 32 synthetic 4 s / 8 kHz two-speaker mixtures
already resident in HBM (BASELINE.json configs[1]).  With N ranks every rank runs its own batch of 32
(weak scaling, configs[2]); there is no collective on the data path - only the timing max-reduce.
Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel = the fused GCFN block (gcfn_fused_kernel, ~1/3 of the step; 56 launches per
                forward with the aux heads): default precision bf16x3 -> bound "mfma" against the dense bf16 MFMA
                peak, achieved = 3 x algorithmic projection FLOPs (the split-fp32 products) / launch time
                measured with hipEvents on the launch stream inside the timed region; the fp32-equivalent rate
                and the HBM-side rate ride along.  With --precision fp32 the dominant kernel is the f32-MFMA
                GCFN up-projection (gemm_kernel<PRO_NORM,EPI_DWGLU,1>) against the 157.3 TFLOP/s f32 peak.
  cpu_baseline  the oracle (CPU restatement of the reference, same aten op sequence; kind "port")
                timed on this host's cores on a bounded sample (B=1, one warm-up + best of 3).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

METRIC = "utterances/sec (4 s, 8 kHz, 2-spk) SepReformer-Base at 1/2/4/8 MI355X"
VARIANT = "SepReformer_Base_WSJ0"
SAMPLES = 32000
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md chip table
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA, same table (2495 TF measured)
HBM_PEAK_GBS = 8000.0                  # HBM3E spec (6.29 TB/s measured copy), same table
GFLOP_PER_UTT_MAIN, GFLOP_PER_UTT_FULL = 164.87, 182.16   # SURVEY.md section 8d (4 s, Base)


def physical_cores() -> int:
    """Distinct (socket, core) pairs from /proc/cpuinfo; falls back to the logical count."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(cfg, max_threads: int):
    """Oracle forward on the host cores, bounded sample: B=1, 4 s.  aten's intra-op threading stops
    scaling (and then regresses) long before a 2-socket host is full for these small ops, so the thread
    count is swept and the FASTEST setting is reported (1 warm-up + best of 2 per setting, ~20-30 s)."""
    from oracle import sepreformer_oracle as orc
    from sepreformer_amd.synth import synth_mixture, synth_state_dict
    sd = synth_state_dict(cfg, 0)
    x = synth_mixture(1, SAMPLES, seed=1234)
    sweep, results = sorted({t for t in (8, 16, 32, 64, max_threads) if t <= max_threads}), {}
    t_budget = time.perf_counter() + 45.0
    with torch.inference_mode():
        for th in sweep:
            if time.perf_counter() > t_budget and results:
                break
            torch.set_num_threads(th)
            orc.model_forward(sd, cfg, x)
            best = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                orc.model_forward(sd, cfg, x)
                best = min(best, time.perf_counter() - t0)
            results[th] = best
    th = min(results, key=results.get)
    return {"value": round(1.0 / results[th], 4), "unit": "utt/s", "cores": th, "kind": "port",
            "sample": f"oracle.model_forward (main + aux heads), B=1 x {SAMPLES} samples, fp32, 1 warm-up + best of 2 "
                      f"per thread count; seconds by threads: " + ", ".join(f"{k}:{v:.2f}" for k, v in results.items())
                      + f"; host has {physical_cores()} physical / {os.cpu_count()} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary heads (NOT the reference's forward)")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the second-precision side measurement")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default=None,
                    help="projection arithmetic (default: the package default / SEPR_PRECISION)")
    args = ap.parse_args()

    from sepreformer_amd import dist as sdist
    from sepreformer_amd import lib as L
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    from sepreformer_amd.synth import synth_mixture

    rank, world, local = sdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the separator path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = L.load()

    cfg = VARIANTS[VARIANT]
    model = Model.from_config(cfg, init_seed=0, precision=args.precision).load_synthetic_(0).eval().to(dev)
    precision = model.precision
    model.compute_aux = not args.no_aux
    B = args.batch
    # each rank separates its own utterances: seeds 1234 + global utterance index
    x = synth_mixture(B, SAMPLES, seed=1234 + rank * B).to(dev)

    def step():
        return model(x)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(dev)

    # parity gate in the same run (rank 0's utterance 0 is the committed golden)
    parity_db = None
    if rank == 0:
        import numpy as np
        from oracle.sepreformer_oracle import agreement_db
        g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_base_4s.npz"))
        if args.warmup == 0:
            out = step()
        main_out = torch.stack([a[0:1] for a in out[0]], 0).cpu()
        parity_db = round(agreement_db(main_out, torch.from_numpy(g["main"])), 1)

    launches_per_step = 56 * 4                  # GCFN launches per forward (56) x sub-batch pipelines; sizes the event pool
    L.check(lib.sepr_prof_start(L.SITE_GCFN_UP, launches_per_step * max(args.steps, 1) + 8), "sepr_prof_start")
    sdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    n_l, ms, fl = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
    L.check(lib.sepr_prof_stop(C.byref(n_l), C.byref(ms), C.byref(fl)), "sepr_prof_stop")
    elapsed = sdist.max_over_ranks(elapsed, dev)

    if rank == 0:
        utt_per_s = world * B * args.steps / elapsed
        gflop = GFLOP_PER_UTT_MAIN if args.no_aux else GFLOP_PER_UTT_FULL
        achieved = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_gcfn_up.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                pmc_rec = json.load(f)
            traffic = pmc_rec.get("hbm_bytes_per_launch")
            ratio = pmc_rec.get("traffic_over_algorithmic")
            if ratio and precision == "bf16x3" and fl.value > 0 and n_l.value > 0:
                # PMC bytes per algorithmic byte (measured on full-batch launches) x this run's algorithmic bytes per launch
                rows_l = fl.value / (18.0 * cfg.feat * cfg.feat + 36.0 * cfg.feat) / n_l.value
                traffic = round(ratio * rows_l * 8.0 * cfg.feat)
        n_launch = max(n_l.value, 1)
        if precision == "fp32":
            dtype = "f32"
            roof = {"kernel": "gemm_kernel<PRO_NORM,EPI_DWGLU,1> (GCFN F->6F projection: LayerNorm prologue, f32 MFMA, "
                              "depthwise-conv+GLU epilogue)",
                    "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic}
        else:
            # the fused GCFN kernel (one launch per GCFN block: LayerNorm, F->6F, depthwise conv + GLU, 3F->F,
            # LayerScale, residual).  Per frame row it moves 8F bytes of HBM (read x, write y, fp32) and issues
            # 3 bf16 MFMA products per algorithmic multiply-add (split-fp32): 3 * 18F^2 * 2 / 8F = 864 bf16 FLOP
            # per HBM byte at F=128, far above the machine balance (2500 TF / 8 TB/s = 312), so the bf16 matrix
            # pipe is the roof.  achieved = bf16 MFMA FLOP/s actually required by the arithmetic (3 x the
            # algorithmic projection FLOPs) over the HIP-event launch time; the fp32-equivalent algorithmic rate
            # and the HBM-side rate are reported next to it.
            F = cfg.feat
            flop_row = 18.0 * F * F + 36.0 * F            # what launch_gcfn_fused reports per row
            rows = fl.value / flop_row
            sec = ms.value / 1e3
            mfma_tf = 3.0 * rows * 18.0 * F * F / 1e12 / sec if sec > 0 else 0.0
            gbs = rows * 8.0 * F / 1e9 / sec if sec > 0 else 0.0
            dtype = "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)"
            roof = {"kernel": "gcfn_fused3_kernel<128,2,4> (and its <128,1,6> instantiation for launches under 17000 rows; whole GCFN block in one launch: LayerNorm, F->6F bf16x3 MFMA, "
                              "depthwise conv k=3 + GLU, 3F->F bf16x3 MFMA, LayerScale, residual)",
                    "bound": "mfma", "achieved": round(mfma_tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(mfma_tf / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(rows / n_launch * 8.0 * F),
                    "algorithmic_fp32_tflops": round(achieved, 2),
                    "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        roof.update({"launches": int(n_l.value), "avg_launch_ms": round(ms.value / n_launch, 4),
                     "algorithmic_gflop_per_launch": round(fl.value / 1e9 / n_launch, 3)})
        if getattr(model, "pipelines", 1) > 1 and B >= 8 * model.pipelines:
            # the timed region runs the batch as `pipelines` half-size sub-batches on separate streams: launches of the
            # same kernel overlap on the device, so each one's event-to-event duration includes the time it shares
            # the CUs with its twin.  The figures of the kernel running ALONE (one pipeline, 2 untimed steps) go
            # next to them; `value` always comes from the timed region above.
            model.pipelines, saved = 1, model.pipelines
            L.check(lib.sepr_prof_start(L.SITE_GCFN_UP, launches_per_step * 2 + 8), "sepr_prof_start")
            step(); step()
            torch.cuda.synchronize(dev)
            n2, ms2, fl2 = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
            L.check(lib.sepr_prof_stop(C.byref(n2), C.byref(ms2), C.byref(fl2)), "sepr_prof_stop")
            model.pipelines = saved
            if ms2.value > 0 and n2.value > 0:
                scale = 3.0 * (18.0 * cfg.feat * cfg.feat) / (18.0 * cfg.feat * cfg.feat + 36.0 * cfg.feat) if precision == "bf16x3" else 1.0
                peak = BF16_MFMA_PEAK_TFLOPS if precision == "bf16x3" else FP32_MFMA_PEAK_TFLOPS
                tf = scale * (fl2.value / 1e12) / (ms2.value / 1e3)
                roof["exclusive"] = {"achieved": round(tf, 1), "frac": round(tf / peak, 4), "launches": int(n2.value),
                                     "avg_launch_ms": round(ms2.value / n2.value, 4),
                                     "note": "same kernel, single pipeline (launches do not share the device)"}
            roof["pipelines"] = saved
        rec = {
            "metric": METRIC, "value": round(utt_per_s, 3), "unit": "utt/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{VARIANT} inference, batch={B} per GPU, 4 s @ 8 kHz, 2 speakers "
                                   f"(BASELINE.json configs[{1 if world == 1 else 2}])",
                       "batch_per_gpu": B, "samples": SAMPLES, "aux_heads": not args.no_aux, "precision": precision,
                       "weights": "synthetic seed 0 (O(1) LayerScale)", "parallelism": f"utterance-sharded x{world}"},
            "parity_db_vs_golden": parity_db,
            # whole-forward algorithmic rate per GPU (fp32-equivalent FLOPs of the reference's op list) and the
            # fraction of the matrix pipe it needs in this arithmetic (bf16x3 issues 3 bf16 MFMA FLOPs per FLOP)
            "model_tflops": round(utt_per_s * gflop / 1e3 / world, 2),
            "model_mfma_frac": round(utt_per_s * gflop / 1e3 / world * (3.0 / BF16_MFMA_PEAK_TFLOPS if precision == "bf16x3"
                                                                         else 1.0 / FP32_MFMA_PEAK_TFLOPS), 4),
            "roofline": roof,
        }
        if world == 1 and not args.no_alt_precision:
            # the other projection arithmetic on the same workload (2 timed steps): exact f32 MFMA vs bf16x3
            alt = "fp32" if precision == "bf16x3" else "bf16x3"
            model.precision = alt
            step(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(2):
                out_alt = step()
            torch.cuda.synchronize(dev)
            alt_s = (time.perf_counter() - t1) / 2
            rec["alt_precision"] = {"precision": alt, "value": round(B / alt_s, 3), "unit": "utt/s",
                                    "ms_per_step": round(1e3 * alt_s, 3),
                                    "parity_db_vs_golden": round(agreement_db(torch.stack([a[0:1] for a in out_alt[0]], 0).cpu(),
                                                                              torch.from_numpy(g["main"])), 1)}
            model.precision = precision
        if world == 1 and not args.no_alt_precision:
            # single-utterance latency: B=1 x 4 s, eager launches vs the captured hipGraph
            x1 = x[:1].contiguous()
            eng = model.engine()

            def med_ms(fn, reps=15):
                ts = []
                for _ in range(reps):
                    torch.cuda.synchronize(dev)
                    t = time.perf_counter()
                    fn()
                    torch.cuda.synchronize(dev)
                    ts.append(time.perf_counter() - t)
                return round(1e3 * sorted(ts)[len(ts) // 2], 3)

            eng.forward(x1, with_aux=True)
            eager = med_ms(lambda: eng.forward(x1, with_aux=True))
            eng.forward_graphed(x1, with_aux=True)
            graphed = med_ms(lambda: eng.forward_graphed(x1, with_aux=True))
            rec["latency_b1"] = {"unit": "ms per 4 s utterance (batch 1, full forward incl. aux heads)",
                                 "eager": eager, "hipgraph": graphed}
        if world == 1 and not args.no_cpu_baseline:
            threads = int(os.environ.get("SEPR_CPU_THREADS", str(physical_cores())))
            rec["cpu_baseline"] = cpu_baseline(cfg, threads)
            rec["speedup_vs_cpu"] = round(utt_per_s / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), flush=True)
    sdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
