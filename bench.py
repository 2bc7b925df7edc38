#!/usr/bin/env python3
"""Throughput of the SepReformer separator on MI355X.

    python bench.py --gpus N --steps K --warmup W         # N > 1 without a launcher: re-executes itself under
                                                          # torch.distributed.run (one rank per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Inference (default, BASELINE.json configs[1] / configs[2]): a "step" is one full ``Model.forward`` (main + the 4
auxiliary heads, exactly what the reference's forward evaluates, model.py:38-54) over one batch of 32 synthetic
4 s / 8 kHz two-speaker mixtures already resident in HBM, followed by the PIT SI-SNR / SI-SNRi metric of the separated
waveforms against the known sources on the device (what the reference's test loop computes per utterance,
engine.py:131) and the 3-scalar RCCL all-reduce of ``[sum SI-SNR, sum SI-SNRi, count]`` - the only collective of the
path (north_star: "RCCL over xGMI for the optional PIT/SI-SNR reduction").  With N ranks every rank runs its own
batch of 32 (weak scaling); there is no collective on the data path.  ``--variant SepReformer_Large_DM_WHAMR``
measures BASELINE configs[3]; ``--mode train`` measures configs[4] (forward + criteria + backward + clip + AdamW,
gradient all-reduce over RCCL).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel = the GCFN block's kernel (fused for Base: ~40 % of the step, 56 launches per forward).
                ``achieved`` = ALGORITHMIC fp32 FLOPs of the launches / their hipEvent durations (measured on the launch
                stream inside the timed region), ``frac`` = achieved / the dense bf16 MFMA peak.  The default
                arithmetic (bf16x3: every fp32 product = 3 bf16 MFMAs) caps that fraction at ``ceiling`` = 1/3;
                ``mfma_pipe_frac`` = 3 x achieved / peak is the share of the matrix pipe actually issued.
                ``traffic`` = HBM bytes per launch from rocprofv3 PMC counters; ``traffic_source`` says where from.
  cpu_baseline  the oracle (CPU restatement of the reference, same aten op sequence; kind "port") timed on this host's
                cores on a bounded sample (B=1 best of 3 per thread count; B=8 once when the time budget allows).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "utterances/sec (4 s, 8 kHz, 2-spk) SepReformer-Base at 1/2/4/8 MI355X"
DEFAULT_VARIANT = "SepReformer_Base_WSJ0"
SAMPLES = 32000
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md chip table
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA, same table (2495 TF measured)
HBM_PEAK_GBS = 8000.0                  # HBM3E spec (6.29 TB/s measured copy), same table
# algorithmic GFLOP per 4 s utterance (main heads, with the 4 aux heads): SURVEY.md section 8d
GFLOP_PER_UTT = {"SepReformer_Base_WSJ0": (164.87, 182.16), "SepReformer_Large_DM_WHAMR": (633.3, 684.14),
                 "SepReformer_Large_DM_WSJ0": (633.3, 684.14), "SepReformer_Large_DM_WHAM": (633.3, 684.14)}


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


def cpu_baseline(cfg, max_threads: int, budget_s: float = 50.0):
    """Oracle forward on the host cores, bounded sample (SURVEY.md section 8d protocol within a time budget): B=1 x 4 s
    with 1 warm-up + best of 3 per thread count (aten's intra-op threading stops scaling long before a 2-socket host
    is full for these small ops, so the count is swept and the FASTEST setting reported), then B=8 once at that count
    if the budget allows."""
    import torch
    from oracle import sepreformer_oracle as orc
    from sepreformer_amd.synth import synth_mixture, synth_state_dict
    sd = synth_state_dict(cfg, 0)
    x = synth_mixture(8, SAMPLES, seed=1234)
    # (all physical cores - 128 here - is 6x slower than 16 threads for these small ops, measured in rounds 1-2: not swept)
    sweep, results = sorted({t for t in (8, 16, 32) if t <= max_threads} or {max_threads}), {}
    t_end = time.perf_counter() + budget_s
    with torch.inference_mode():
        for th in sweep:
            if time.perf_counter() > t_end - 8.0 and results:
                break
            torch.set_num_threads(th)
            orc.model_forward(sd, cfg, x[:1])
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                orc.model_forward(sd, cfg, x[:1])
                best = min(best, time.perf_counter() - t0)
            results[th] = best
        th = min(results, key=results.get)
        b8 = None
        if time.perf_counter() + 16.0 * results[th] < t_end:      # a batch of 8 takes ~2x 8 single forwards
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            orc.model_forward(sd, cfg, x)
            b8 = time.perf_counter() - t0
    over_ref = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "PINNING.json")) as f:
            def find(o):
                if isinstance(o, dict):
                    if "oracle_over_ref" in o:
                        return o["oracle_over_ref"]
                    for v in o.values():
                        r = find(v)
                        if r is not None:
                            return r
                return None
            over_ref = find(json.load(f))
    except (OSError, ValueError):
        pass
    rec = {"value": round(1.0 / results[th], 4), "unit": "utt/s", "cores": th, "kind": "port",
           "sample": f"oracle.model_forward (main + aux heads), fp32; B=1 x {SAMPLES} samples: 1 warm-up + best of 3 per "
                     f"thread count, seconds by threads: " + ", ".join(f"{k}:{v:.2f}" for k, v in results.items())
                     + (f"; B=8 once at {th} threads: {b8:.2f} s = {8.0 / b8:.3f} utt/s" if b8 else "; B=8 skipped (time budget)")
                     + f"; host has {physical_cores()} physical / {os.cpu_count()} logical cores"
                     + (f"; port = the oracle, which takes {over_ref:.2f}x the imported reference's wall time on the same inputs in the "
                        f"build container (tests/golden/PINNING.json oracle_over_ref)" if over_ref else "")}
    if over_ref:
        rec["oracle_over_ref_walltime"] = round(float(over_ref), 3)
    if b8:
        rec["value_b8"] = round(8.0 / b8, 4)
    return rec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default 32; 8 in train mode)")
    ap.add_argument("--variant", default=DEFAULT_VARIANT, help="model variant (models/<variant>/configs.yaml)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary heads (NOT the reference's forward)")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the second-precision / latency side measurements")
    ap.add_argument("--no-metric", action="store_true", help="skip the per-step PIT SI-SNR metric + RCCL reduction")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks use GPU 0 and the gloo backend (exercises the N > 1 path end to end on a 1-GPU box; "
                         "the throughput it prints is NOT a scaling number)")
    ap.add_argument("--train-graphs", choices=["step", "split", "off"], default=None,
                    help="train mode: step (default) = the whole step as hipGraph replays, split = separator forward / backward graphs with eager "
                         "criteria / clip / optimizer, off = eager")
    ap.add_argument("--no-train-graphs", action="store_true", help="train mode: launch every kernel from the host instead of replaying captured hipGraphs")
    ap.add_argument("--pmc", choices=["auto", "on", "off"], default="auto",
                    help="HBM traffic of the dominant kernel from rocprofv3 PMC passes of a short sub-run (auto: in the default 1-GPU run)")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default=None,
                    help="projection arithmetic (default: the package default / SEPR_PRECISION)")
    return ap.parse_args()


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: run N ranks of this script under torch.distributed.run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


PMC_KERNEL_RE = "gcfn_fused3_kernel<[0-9]+, [0-9], [0-9], 0, false, false, [0-9]>"     # the GCFN instantiations of the fused kernel (not the plain GLU-MLP mode)


LARGE_KERNEL_RE = "gcfn_fused3_kernel<256, 2, 4, 0, false, false, 0>"       # Large: the fused F = 256 GCFN (round 6; before: gemm_x3w?_kernel<1, 7, 1>)
TN_KERNEL_RE = "gemm_tnd?_kernel"    # every instantiation of the weight-gradient contraction, register-staged and LDS-DMA (the training line's roofline kernel)


def measure_pmc_traffic(budget_s=90.0, kernel_re=None, sub_args=None):
    """HBM bytes per launch of the fused GCFN kernel, measured NOW: ``rocprofv3 --pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` in separate
    passes (kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short sub-run of this script at the same batch.  Returns
    (record, None) or (None, reason).  FETCH_SIZE x2: the guide's gfx950 correction for 16 B/lane streaming reads; counters are KiB."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sums, launches, algo = {}, 0, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"sepr_pmc_{ctr}_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "--kernel-include-regex", kernel_re or PMC_KERNEL_RE, "--output-format", "csv", "-d", d, "-o", "t", "--",
               sys.executable, os.path.abspath(__file__)] + (sub_args or ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-precision", "--pmc", "off"])
        env = dict(os.environ, TMPDIR="/tmp", SEPR_PIPELINES="1")      # one pipeline: every counted launch has the full-batch size
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            out = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=budget_s, text=True)
        except (subprocess.TimeoutExpired, OSError) as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"{ctr} pass: {type(e).__name__}"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if out.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"{ctr} pass: rc {out.returncode}, {len(files)} counter files"
        with open(files[0]) as f:
            rows = [r for r in csv.DictReader(f) if r.get("Counter_Name") == ctr]
        shutil.rmtree(d, ignore_errors=True)
        if not rows:
            return None, f"{ctr} pass: no rows for {kernel_re or PMC_KERNEL_RE}"
        sums[ctr], launches = sum(float(r["Counter_Value"]) for r in rows), len(rows)
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                try:
                    algo = json.loads(line)["roofline"].get("algorithmic_bytes_per_launch")
                except (ValueError, KeyError):
                    pass
    fetch = 2.0 * 1024.0 * sums["FETCH_SIZE"] / launches
    write = 1024.0 * sums["WRITE_SIZE"] / launches
    return {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
            "launches": launches, "algorithmic_bytes_per_launch": algo,
            "traffic_over_algorithmic": round((fetch + write) / algo, 3) if algo else None}, None


TRAIN_SUBS = (("bf16x3", "bf16x3", 16), ("bf16", "bf16", 16), ("bf16_b32", "bf16", 32))      # training sub-records: (name, precision, batch)
SUB_STEPS_LARGE = 10       # timed steps of the Large sub-record (round 4: 3) and of every training sub-record (round 4: 2): the default run
SUB_STEPS_TRAIN = 8        # uses a few minutes of its budget, and 2-step records could not be told from box-to-box noise
PARITY_MIN_DB = 80.0        # agreement with the golden waveform (SURVEY.md section 8d gate i)
PIT_GATE_DB = 1e-3          # |PIT SI-SNR(hip) - PIT SI-SNR(reference)| per utterance (north_star; gate ii)
GOLDEN_4S = {"SepReformer_Base_WSJ0": "e2e_base_4s.npz", "SepReformer_Large_DM_WHAMR": "e2e_large_whamr_4s.npz"}


def measure_infer(args, variant, steps, warmup, rank, world, dev, lib, full):
    """One inference measurement (``full``: the headline line with every side measurement; else a bounded sub-record of
    another BASELINE configuration inside the default run)."""
    import torch
    from sepreformer_amd import dist as sdist
    from sepreformer_amd import lib as L
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.criterion import pit_sisnr
    from sepreformer_amd.model import Model
    from sepreformer_amd.synth import synth_sources

    cfg = VARIANTS[variant]
    if args.precision == "bf16":
        raise SystemExit("--precision bf16 is a TRAINING arithmetic (plain bf16 operands do not pass the 1e-3 dB SI-SNR gate): "
                         "use it with --mode train; inference runs fp32 or bf16x3")
    model = Model.from_config(cfg, init_seed=0, precision=args.precision).load_synthetic_(0).eval().to(dev)
    precision = model.precision
    model.compute_aux = not args.no_aux
    B = args.batch or 32
    # each rank separates its own utterances: seeds 1234 + global utterance index
    src = torch.from_numpy(synth_sources(B, SAMPLES, seed=1234 + rank * B)).to(dev)        # [B, 2, T]
    x = src.sum(1).contiguous()
    tgt = src.permute(1, 0, 2).contiguous()                                                 # [S, B, T]
    metric_acc = torch.zeros(3, dtype=torch.float64, device=dev)
    count = torch.tensor(float(B), dtype=torch.float64, device=dev)

    def step():
        out = model(x)
        if not args.no_metric:
            # the reference's test-loop metric on the device (engine.py:131) + the path's only collective
            est = torch.stack(out[0], 0)
            m = pit_sisnr(est, tgt[..., : est.shape[-1]], mixture=x[..., : est.shape[-1]])
            acc = torch.stack([-m["loss"].double().sum(), m["sisnri"].double().sum(), count])
            metric_acc.copy_(sdist.reduce_metric_sums(acc))
        return out

    # Throughput mode: the batch as two half-batch pipelines on two streams (engine.forward_split: the pipelines fill each other's
    # launch tails; bit-identical results, +2.5-3.7 % at batch 32 in round 3).  Two concurrent launches share the device, so the
    # per-launch event durations of the roofline are taken from a second, single-pipeline region right after the timed one.
    # Round 4: that IS the product default now (Model.pipelines = 0 = auto: two pipelines from 16 utterances up), so the timed region
    # runs the public call in its default mode.
    if args.share_gpu:
        model.pipelines = 1
    pl_setting = model.pipelines
    pl = model.effective_pipelines(B)
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize(dev)

    launches_per_step = 56 * 4                  # GCFN launches per forward (56) x sub-batch pipelines; sizes the event pool
    if pl == 1:
        L.check(lib.sepr_prof_start(L.SITE_GCFN_UP, launches_per_step * max(steps, 1) + 8), "sepr_prof_start")
    sdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    n_l, ms, fl = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
    single = None
    if pl == 1:
        L.check(lib.sepr_prof_stop(C.byref(n_l), C.byref(ms), C.byref(fl)), "sepr_prof_stop")
    else:
        model.pipelines = 1
        step()
        torch.cuda.synchronize(dev)
        L.check(lib.sepr_prof_start(L.SITE_GCFN_UP, launches_per_step * max(steps, 1) + 8), "sepr_prof_start")
        sdist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        sdist.barrier()
        e1 = sdist.max_over_ranks(time.perf_counter() - t1, dev)
        L.check(lib.sepr_prof_stop(C.byref(n_l), C.byref(ms), C.byref(fl)), "sepr_prof_stop")
        single = {"value": round(world * B * steps / e1, 3), "unit": "utt/s", "ms_per_step": round(1e3 * e1 / max(steps, 1), 3), "steps": steps}
        model.pipelines = pl_setting
    # per-rank spread of the timed region BEFORE the max: the first multi-GPU run shows its launch jitter directly (min == max on one rank)
    rank_min_s, rank_max_s = sdist.min_max_over_ranks(elapsed, dev)
    elapsed = sdist.max_over_ranks(elapsed, dev)
    # the path's collective on its own, EVERY rank takes part: wall time of one synchronised 3-scalar all-reduce, median of 10
    ar_ms = None
    if not args.no_metric:
        ts, probe = [], torch.zeros(3, dtype=torch.float64, device=dev)
        for _ in range(12):
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            sdist.reduce_metric_sums(probe)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - ta)
        ar_ms = 1e3 * sorted(ts[2:])[5]
    rccl_ranks = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None

    # parity gates in the same run (rank 0): utterance 0 against the committed golden waveform of this variant, and the
    # north_star gate |PIT SI-SNR(hip) - PIT SI-SNR(reference)| per utterance (reference values: tests/golden)
    parity_db = pit_delta = None
    g = None
    if rank == 0 and variant in GOLDEN_4S and os.path.exists(os.path.join(ROOT, "tests", "golden", GOLDEN_4S[variant])):
        import numpy as np
        from oracle.sepreformer_oracle import agreement_db, pit_si_snr_db
        g = np.load(os.path.join(ROOT, "tests", "golden", GOLDEN_4S[variant]))
        # (after the timed region: the host-side gate arithmetic idles the GPU for ~0.5 s, which would otherwise put the first
        #  timed steps on ramping clocks.  model(x), NOT step(): step() contains the all-reduce and only rank 0 is here)
        out = model(x)
        main_out = torch.stack([a[0:1] for a in out[0]], 0).cpu()
        parity_db = round(agreement_db(main_out, torch.from_numpy(g["main"])), 1)
        T_ = out[0][0].shape[-1]
        gate = os.path.join(ROOT, "tests", "golden", "pit_gate_base_b32.npz")
        if variant == DEFAULT_VARIANT and os.path.exists(gate):
            ref_pit = np.load(gate)["ref_pit_db"]
        else:                                   # one golden utterance: the reference's own separated waveforms of utterance 0
            ref_pit = pit_si_snr_db([torch.from_numpy(g["main"][s]) for s in range(cfg.num_spks)],
                                    [src[:1, s, :T_].cpu() for s in range(cfg.num_spks)]).numpy()
        nb = min(B, len(ref_pit))
        got = pit_si_snr_db([a[:nb].cpu() for a in out[0]], [src[:nb, s, :T_].cpu() for s in range(cfg.num_spks)])
        pit_delta = float((got - torch.from_numpy(np.asarray(ref_pit[:nb]))).abs().max())
    if rank != 0:
        return None

    F = cfg.feat
    utt_per_s = world * B * steps / elapsed
    gflop = GFLOP_PER_UTT.get(variant, (0.0, 0.0))[0 if args.no_aux else 1]
    n_launch = max(n_l.value, 1)
    sec = ms.value / 1e3
    algo_tf = (fl.value / 1e12) / sec if sec > 0 else 0.0            # algorithmic fp32 FLOPs / launch time
    fused = precision == "bf16x3" and F in (64, 128, 256) and os.environ.get("SEPR_FUSE_GCFN", "1") != "0" and not (F == 256 and os.environ.get("SEPR_FUSE_GCFN256", "1") == "0")
    # rows per launch: the fused kernel reports 18F^2 + 36F FLOPs per row (both projections + the conv), the generic
    # up-projection 2 * 6F * F per row
    rows = fl.value / (18.0 * F * F + 36.0 * F) if fused else fl.value / (12.0 * F * F)
    # algorithmic HBM bytes per row of the profiled launch: fused block = x in + y out (8F); generic up-projection = x in + the gated
    # tensor out (4F + 12F)
    bytes_per_row = 8.0 * F if fused else 16.0 * F
    algo_bytes_launch = rows / n_launch * bytes_per_row
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_gcfn_up.json")
    if fused and variant == DEFAULT_VARIANT and os.path.exists(pmc):
        with open(pmc) as f:
            pmc_rec = json.load(f)
        ratio = pmc_rec.get("traffic_over_algorithmic")
        if ratio and algo_bytes_launch:
            # PMC bytes per algorithmic byte (rocprofv3 FETCH_SIZE / WRITE_SIZE passes on full-batch launches,
            # tools/pmc_traffic.sh) x this run's algorithmic bytes per launch
            traffic = round(ratio * algo_bytes_launch)
            traffic_src = ("static: profiles/pmc_gcfn_up.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                           f"tools/pmc_traffic.sh, collected {pmc_rec.get('collected', 'in round 1')}); not re-measured in this run")
    if precision == "fp32":
        dtype = "f32"
        peak, mult, ceiling = FP32_MFMA_PEAK_TFLOPS, 1.0, 1.0
        kern = ("gemm_kernel<PRO_NORM,EPI_DWGLU,1> (GCFN F->6F projection: LayerNorm prologue, f32 MFMA, "
                "depthwise-conv+GLU epilogue)")
    else:
        mult = 3.0
        peak, ceiling = BF16_MFMA_PEAK_TFLOPS, 1.0 / mult
        dtype = "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)"
        kern = (("gcfn_fused3_kernel<256,2,4> (round 6: the row-stationary GCFN at F = 256 in the one-wave-per-SIMD regime - four 30-frame waves with 512 "
                 "registers each, one 106 KB workgroup per CU, inline-asm LDS-DMA weight chunks; whole GCFN block in one launch)" if F == 256 else
                 "gcfn_fused3_kernel<F,2,4> (and its ring-form <F,1,4..6,..,3> instantiations for launches of at most one tile per CU; whole GCFN block in "
                 "one launch: LayerNorm, F->6F MFMA, depthwise conv k=3 + GLU, 3F->F MFMA, LayerScale, residual)")
                if fused else
                "gemm_x3w_kernel<PRO_NORM,EPI_DWGLU,1> (GCFN F->6F projection of the generic path on the 128x256 wide core - "
                "gemm_x3_kernel for launches too small for it: LayerNorm prologue, bf16x3 MFMA, depthwise-conv+GLU epilogue)")
    # matrix FLOPs only (the conv's 36F per row ride on the VALU) for the pipe-occupancy figure
    mfma_tf = mult * (rows * 18.0 * F * F if fused else fl.value) / 1e12 / sec if sec > 0 else 0.0
    roof = {"kernel": kern, "bound": "mfma", "achieved": round(algo_tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(algo_tf / peak, 4), "frac_algorithmic": round(algo_tf / peak, 4),
            "mfma_pipe_frac": round(mfma_tf / peak, 4), "ceiling": round(ceiling, 4),
            "frac_of_ceiling": round(algo_tf / peak / ceiling, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "launches": int(n_l.value), "avg_launch_ms": round(ms.value / n_launch, 4),
            "algorithmic_gflop_per_launch": round(fl.value / 1e9 / n_launch, 3),
            "measured_on": ("the timed steps" if pl == 1 else
                            f"{steps} steps of the same workload run as ONE pipeline right after the timed region (two concurrent pipelines share the "
                            "device: per-launch durations inside the timed region would not describe one kernel); rocprofv3 summary: the same "
                            "command with SEPR_PIPELINES=1")}
    if algo_bytes_launch:
        gbs = rows * bytes_per_row / 1e9 / sec if sec > 0 else 0.0
        roof.update({"algorithmic_bytes_per_launch": round(algo_bytes_launch), "hbm_gbs": round(gbs, 1),
                     "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
    cfg_idx = {DEFAULT_VARIANT: 1 if world == 1 else 2, "SepReformer_Large_DM_WHAMR": 3}.get(variant)
    acc = metric_acc.cpu()
    rec = {
        "metric": METRIC if variant == DEFAULT_VARIANT else METRIC.replace("SepReformer-Base", variant),
        "value": round(utt_per_s, 3), "unit": "utt/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / max(steps, 1), 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{variant} inference, batch={B} per GPU, 4 s @ 8 kHz, 2 speakers"
                               + (f" (BASELINE.json configs[{cfg_idx}])" if cfg_idx else ""),
                   "batch_per_gpu": B, "samples": SAMPLES, "aux_heads": not args.no_aux, "precision": precision,
                   "pipelines": pl, "pipelines_source": "Model default (auto: 2 from 16 utterances up)" if pl_setting == 0 else "SEPR_PIPELINES / --share-gpu",
                   "weights": "synthetic seed 0 (O(1) LayerScale)",
                   "parallelism": f"utterance-sharded x{world}" + (" (DEBUG: all ranks share GPU 0, gloo collective)" if args.share_gpu else ""),
                   "step": "Model.forward (main + aux heads)" + ("" if args.no_metric else
                           " + device PIT SI-SNR/SI-SNRi of the batch + 3-scalar all-reduce")},
        "parity_db_vs_golden": parity_db,
        "pit_si_snr_max_abs_delta_db": None if pit_delta is None else float(f"{pit_delta:.3e}"),
        # the gates of SURVEY.md section 8d, enforced: a failed gate makes the process exit non-zero (main())
        "parity_ok": None if parity_db is None else bool(parity_db >= PARITY_MIN_DB and pit_delta is not None and pit_delta <= PIT_GATE_DB),
        "rccl_ranks": rccl_ranks, "collective_backend": backend,
        "reduced_metric": None if args.no_metric else {
            "utterances": int(acc[2].item()), "mean_pit_si_snr_db": round(float(acc[0] / acc[2]) / cfg.num_spks, 4),
            "mean_si_snri_db": round(float(acc[1] / acc[2]) / cfg.num_spks, 4),
            "note": "sum over ranks of the last step's per-utterance metric (random weights: values are not a quality claim)"},
        # whole-forward algorithmic rate per GPU (fp32-equivalent FLOPs of the reference's op list) and the
        # fraction of the matrix pipe it needs in this arithmetic
        "model_tflops": round(utt_per_s * gflop / 1e3 / world, 2),
        "model_frac_algorithmic": round(utt_per_s * gflop / 1e3 / world / peak, 4),
        "model_mfma_frac": round(utt_per_s * gflop / 1e3 / world * mult / peak, 4),
        "roofline": roof,
        # measured, every run: the ranks' own wall times of the timed region (ms per step, before the max-over-ranks) and the metric
        # all-reduce on its own - on N > 1 GPUs the min/max spread is the host-side launch jitter dp8_prediction does not model
        "per_rank_ms_per_step": {"min": round(1e3 * rank_min_s / max(steps, 1), 3), "max": round(1e3 * rank_max_s / max(steps, 1), 3), "ranks": world},
        "metric_allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
    }
    if single is not None:
        rec["single_pipeline"] = single
    if not args.no_metric and variant == DEFAULT_VARIANT:
        # configs[2] (batch 256 over 8 GPUs = this per-rank workload x 8) from measured pieces: the data path has no collective, the
        # per-step exchange is the 24-byte metric all-reduce (latency-bound: a few tens of microseconds over xGMI against a >= 24 ms step)
        step_ms = 1e3 * elapsed / max(steps, 1)
        lat8 = max(ar_ms, 0.06)                  # <= ~60 us for an 8-rank latency-bound all-reduce over xGMI
        rec["dp8_prediction"] = {"n_gpus": 8, "per_rank_step_ms_measured": round(step_ms, 3), "metric_allreduce_bytes": 24,
                                 "metric_allreduce_ms_measured_here": round(ar_ms, 4), "allreduce_ranks_here": rccl_ranks,
                                 "metric_allreduce_ms_8rank_model": round(lat8, 4),
                                 "predicted_step_ms": round(step_ms - (ar_ms if rccl_ranks == 1 else 0.0) + lat8, 3),
                                 "predicted_utt_per_s": round(8 * B / (step_ms - (ar_ms if rccl_ranks == 1 else 0.0) + lat8) * 1e3, 1),
                                 "predicted_scaling_efficiency": round(step_ms / (step_ms - (ar_ms if rccl_ranks == 1 else 0.0) + lat8), 4),
                                 "note": "prediction, not a measurement (no multi-GPU node was available to the builder): utterance shards are independent, "
                                         "weights replicated once, the only collective is this 3-scalar all-reduce; what a real run adds is host-side launch "
                                         "jitter across 8 processes (each rank drives its own ~450-launch forward)"}
    if not full:
        del model
        torch.cuda.empty_cache()
        return rec
    if world == 1 and not args.no_alt_precision and g is not None:
        from oracle.sepreformer_oracle import agreement_db
        # the other projection arithmetic on the same workload (2 timed steps): exact f32 MFMA vs bf16x3
        alt = "fp32" if precision != "fp32" else "bf16x3"
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
        rec["alt_precision"]["parity_ok"] = bool(rec["alt_precision"]["parity_db_vs_golden"] >= PARITY_MIN_DB)
        model.precision = precision
    if world == 1 and not args.no_alt_precision:
        # single-utterance latency: B=1 x 4 s through the PUBLIC call, model(x)
        x1 = x[:1].contiguous()

        def med_ms(fn, reps=15):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                fn()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t)
            return round(1e3 * sorted(ts)[len(ts) // 2], 3)

        model(x1)
        eager = med_ms(lambda: model(x1))
        model.use_graphs = True
        try:
            model(x1)
            graphed = med_ms(lambda: model(x1))
        finally:
            model.use_graphs = False
        rec["latency_b1"] = {"unit": "ms per 4 s utterance (batch 1, model(x): full forward incl. aux heads)",
                             "eager": eager, "hipgraph": graphed,
                             # (the launch count of a batch-1 forward is a rocprofv3 kernel trace, not re-counted here: profiles/r06_b1_launches.txt)
                             "note": "round 6: launches with at most one tile per CU run the latency form of the fused GCFN / GLU-MLP kernels (3-stage LDS weight "
                                     "ring, inline-asm LDS-DMA, one barrier per chunk, 4 waves); the latency is still the SUM of per-kernel latencies (hipGraph replay is no faster)"}
    del model
    torch.cuda.empty_cache()
    return rec


def gate_failures(rec) -> list:
    """Every parity gate carried by the line (headline, alt precision, sub-records); a non-empty list makes bench.py exit 3
    AFTER printing the line, so that the driver's return code reflects a wrong answer."""
    bad = []
    if rec.get("parity_ok") is False:
        bad.append(f"headline: {rec.get('parity_db_vs_golden')} dB vs golden (>= {PARITY_MIN_DB}), PIT delta {rec.get('pit_si_snr_max_abs_delta_db')} dB (<= {PIT_GATE_DB})")
    if (rec.get("alt_precision") or {}).get("parity_ok") is False:
        bad.append(f"alt_precision: {rec['alt_precision'].get('parity_db_vs_golden')} dB vs golden")
    if (rec.get("large") or {}).get("parity_ok") is False:
        bad.append(f"large: {rec['large'].get('parity_db_vs_golden')} dB, PIT delta {rec['large'].get('pit_si_snr_max_abs_delta_db')} dB")
    for name, tr in (rec.get("train") or {}).items():
        if isinstance(tr, dict) and tr.get("parity_ok") is False:
            bad.append(f"train.{name}: {tr.get('parity_note')}")
    return bad


def make_summary(rec) -> dict:
    """Every headline number of the line in <= 1500 characters, emitted as the LAST key: a consumer that keeps only the tail of the
    ~15 KB line still sees all of them (value / ms per step / fraction of the relevant peak / parity of each record)."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    roof = rec.get("roofline") or {}
    out = {"utt_s": rec.get("value"), "ms": rec.get("ms_per_step"), "parity_ok": rec.get("parity_ok"), "db": rec.get("parity_db_vs_golden"),
           "pit_d": rec.get("pit_si_snr_max_abs_delta_db"), "roof": [roof.get("frac"), roof.get("avg_launch_ms"), roof.get("traffic_over_algorithmic")],
           "one_pipe": g(rec, "single_pipeline", "value"), "fp32": [g(rec, "alt_precision", "value"), g(rec, "alt_precision", "parity_db_vs_golden")],
           "lat_b1_ms": [g(rec, "latency_b1", "eager"), g(rec, "latency_b1", "hipgraph")],
           "cpu": [g(rec, "cpu_baseline", "value"), g(rec, "cpu_baseline", "cores"), rec.get("speedup_vs_cpu")]}
    lg = rec.get("large")
    if isinstance(lg, dict):
        out["large"] = ({"err": lg["error"][:80]} if "error" in lg else
                        {"utt_s": lg.get("value"), "ms": lg.get("ms_per_step"), "steps": lg.get("steps"), "ok": lg.get("parity_ok"), "db": lg.get("parity_db_vs_golden"),
                         "pit_d": lg.get("pit_si_snr_max_abs_delta_db"), "roof": [g(lg, "roofline", "frac"), g(lg, "roofline", "traffic_over_algorithmic")],
                         "cpu": g(lg, "cpu_baseline", "value")})
    tr = rec.get("train")
    if isinstance(tr, dict):
        out["train"] = {}
        for name, t in tr.items():
            if not isinstance(t, dict):
                continue
            out["train"][name] = ({"err": t["error"][:80]} if "error" in t else
                                  {"utt_s": t.get("value"), "ms": t.get("ms_per_step"), "steps": t.get("steps"), "frac_alg": t.get("model_frac_algorithmic"),
                                   "tn_hbm": [g(t, "roofline", "frac"), g(t, "roofline", "traffic_over_algorithmic")],
                                   "mid": [g(t, "roofline_gcfn_bwd", "avg_launch_ms"), g(t, "roofline_gcfn_bwd", "frac")],
                                   "loss": t.get("loss"), "gn": t.get("grad_norm"), "att": t.get("attempts", 1)})
    if rec.get("gate_failures"):
        out["gate_failures"] = len(rec["gate_failures"])
    return out


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner from C
    when a communicator is created), so file descriptor 1 is pointed at stderr for the rest of the process and the JSON line
    goes to a private duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(rec):
        os.write(real, (json.dumps(rec) + "\n").encode())

    return emit


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    emit = protect_stdout()
    args._emit = emit
    if args.mode == "train":
        from sepreformer_amd import train_bench
        return train_bench.main(args)

    import torch
    from sepreformer_amd import dist as sdist
    from sepreformer_amd import lib as L
    from sepreformer_amd.config import VARIANTS

    torch.manual_seed(0)        # every seed a model derives from torch.initial_seed() (dropout) is then the same on every box
    # a process group also at world size 1: the metric reduction then really runs through RCCL on a 1-GPU box
    dist_err = None
    try:
        rank, world, local = sdist.init_from_env("gloo" if args.share_gpu else None, single_rank_group=not args.share_gpu)
    except Exception as e:                      # noqa: BLE001 - a box whose RCCL cannot start still gets measured (and says so)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise
        dist_err = f"{type(e).__name__}: {e}"[:300]
        rank, world, local = 0, 1, 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the separator path)")
    if args.share_gpu:
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = L.load()

    variant = args.variant
    rec = measure_infer(args, variant, args.steps, args.warmup, rank, world, dev, lib, full=True)
    if rank == 0:
        if dist_err:
            rec["collective_error"] = dist_err
        default_run = world == 1 and variant == DEFAULT_VARIANT and not args.no_alt_precision and args.batch is None
        if default_run:
            # The other single-GPU configurations of BASELINE.json as bounded sub-records of the SAME driver-observed line:
            # configs[3] (Large_DM_WHAMR inference, with its own parity gate) and configs[4] (the training step).
            t_sub = time.perf_counter()
            try:
                sub = measure_infer(args, "SepReformer_Large_DM_WHAMR", SUB_STEPS_LARGE, 2, rank, world, dev, lib, full=False)
                rec["large"] = {k: sub[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "parity_db_vs_golden",
                                                     "pit_si_snr_max_abs_delta_db", "parity_ok", "model_tflops", "model_frac_algorithmic",
                                                     "model_mfma_frac", "roofline")}
            except Exception as e:              # noqa: BLE001
                rec["large"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if "error" not in rec["large"] and args.pmc != "off" and rec["large"]["roofline"].get("algorithmic_bytes_per_launch"):
                # HBM traffic of Large's dominant kernel (the GCFN up-projection with the conv + GLU epilogue), measured in this run
                try:
                    pm, why = measure_pmc_traffic(90.0, LARGE_KERNEL_RE, ["--variant", "SepReformer_Large_DM_WHAMR", "--steps", "1", "--warmup", "1",
                                                                         "--no-cpu-baseline", "--no-alt-precision", "--pmc", "off"])
                except Exception as e:          # noqa: BLE001
                    pm, why = None, f"{type(e).__name__}: {e}"[:200]
                roof = rec["large"]["roofline"]
                if pm and pm.get("traffic_over_algorithmic"):
                    roof["traffic"] = round(pm["traffic_over_algorithmic"] * roof["algorithmic_bytes_per_launch"])
                    roof["traffic_over_algorithmic"] = pm["traffic_over_algorithmic"]
                    roof["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over {pm['launches']} "
                                              f"launches of a 1-step Large sub-run; fetch {pm['fetch_bytes_per_launch']} + write {pm['write_bytes_per_launch']} B per launch")
                else:
                    roof["traffic_source"] = f"live PMC passes failed: {why}"
            if "error" not in rec["large"] and not args.no_cpu_baseline:
                # the oracle on the host for this variant: ONE forward of one utterance, no warm-up (a Large forward takes seconds)
                try:
                    from oracle import sepreformer_oracle as orc
                    from sepreformer_amd.synth import synth_mixture, synth_state_dict
                    lcfg = VARIANTS["SepReformer_Large_DM_WHAMR"]
                    sd_l = synth_state_dict(lcfg, 0)
                    x1 = synth_mixture(1, SAMPLES, seed=1234)
                    torch.set_num_threads(16)
                    with torch.inference_mode():
                        tc = time.perf_counter()
                        orc.model_forward(sd_l, lcfg, x1)
                        tc = time.perf_counter() - tc
                    rec["large"]["cpu_baseline"] = {"value": round(1.0 / tc, 4), "unit": "utt/s", "cores": 16, "kind": "port",
                                                    "sample": f"oracle.model_forward, fp32, B=1 x {SAMPLES} samples, ONE forward without warm-up: {tc:.2f} s at 16 threads"}
                    del sd_l
                except Exception as e:          # noqa: BLE001
                    rec["large"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            # the training lines run in their own processes: a fault of the training path cannot take the headline line with it
            rec["train"] = {}
            keys = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "capture_fallback", "host_enqueue_ms_per_step",
                    "host_loop_ms_per_step", "loss", "grad_norm", "collective_backend", "allreduce_bytes_per_step", "model_tflops",
                    "model_frac_algorithmic", "dp8_prediction", "roofline", "roofline_gcfn_bwd")
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            for name, prec, tb in TRAIN_SUBS:
                try:
                    # (one retry: a sub-run has died in the RCCL watchdog thread about once in 15 runs on the 1-GPU boxes - rc -6 before its
                    #  first step, never reproduced in isolation; the stderr of a failed attempt is kept, `attempts` says what happened)
                    for attempt in (1, 2):
                        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", "train", "--batch", str(tb), "--steps", str(SUB_STEPS_TRAIN), "--warmup", "2",
                                              "--precision", prec], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420, text=True)
                        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                        if out.returncode == 0 and lines:
                            break
                        try:
                            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                            with open(os.path.join(ROOT, "gpurun_out", f"bench_train_{name}.attempt{attempt}.stderr"), "w") as f:
                                f.write(out.stderr)
                        except OSError:
                            pass
                    if out.returncode != 0 or not lines:
                        try:                    # the whole stderr of a failed sub-run is worth keeping (gpurun_out/ travels back)
                            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                            with open(os.path.join(ROOT, "gpurun_out", f"bench_train_{name}.stderr"), "w") as f:
                                f.write(out.stderr)
                        except OSError:
                            pass
                        head = [ln for ln in out.stderr.splitlines() if "Error" in ln or "error" in ln or "terminate" in ln][:3]
                        raise RuntimeError(f"rc {out.returncode}: {' | '.join(head)[:400]} ... {out.stderr[-200:]}")
                    tr = json.loads(lines[-1])
                    rec["train"][name] = {k: tr.get(k) for k in keys}
                    if attempt > 1:
                        rec["train"][name]["attempts"] = attempt
                except Exception as e:          # noqa: BLE001
                    rec["train"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            # the training rooflines' HBM traffic measured in THIS run as well, for every training record (the contraction kernels of one eager
            # step at the record's own batch and arithmetic)
            if args.pmc != "off":
                t_pmc = time.perf_counter()
                for name, prec, tb in TRAIN_SUBS:
                    trn = rec["train"].get(name) or {}
                    roof = trn.get("roofline")
                    if not (isinstance(roof, dict) and roof.get("algorithmic_bytes_per_launch")):
                        continue
                    try:
                        pm, why = measure_pmc_traffic(150.0, TN_KERNEL_RE, ["--mode", "train", "--batch", str(tb), "--steps", "1", "--warmup", "0", "--precision", prec,
                                                                           "--train-graphs", "off"])
                    except Exception as e:      # noqa: BLE001
                        pm, why = None, f"{type(e).__name__}: {e}"[:200]
                    if pm and pm.get("traffic_over_algorithmic"):
                        roof["traffic"] = round(pm["traffic_over_algorithmic"] * roof["algorithmic_bytes_per_launch"])
                        roof["traffic_over_algorithmic"] = pm["traffic_over_algorithmic"]
                        roof["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over the "
                                                  f"{pm['launches']} gemm_tn / gemm_tnd launches of an eager sub-run at the same batch and arithmetic; FETCH_SIZE x2 (gfx950 correction of "
                                                  f"MI355X_MICROARCH.md), KiB units; fetch {pm['fetch_bytes_per_launch']} + write {pm['write_bytes_per_launch']} B per launch "
                                                  f"there = {pm['traffic_over_algorithmic']} x the algorithmic operand bytes (the write side is the split-M partial tiles)")
                    else:
                        roof["traffic"] = None
                        roof["traffic_source"] = f"live PMC passes failed: {why}"
                rec["train_pmc_s"] = round(time.perf_counter() - t_pmc, 1)
            rec["sub_records_s"] = round(time.perf_counter() - t_sub, 1)
        if world == 1 and (args.pmc == "on" or (args.pmc == "auto" and default_run)) and rec["roofline"].get("algorithmic_bytes_per_launch"):
            # traffic re-measured in THIS run (two short PMC sub-runs at the same batch: every launch they count has the timed launches' size)
            t_pmc = time.perf_counter()
            torch.cuda.synchronize(dev)
            try:
                pm, why = measure_pmc_traffic()
            except Exception as e:              # noqa: BLE001 - the side measurement must never cost the line
                pm, why = None, f"{type(e).__name__}: {e}"[:200]
            if pm and pm.get("traffic_over_algorithmic"):
                rec["roofline"]["traffic"] = round(pm["traffic_over_algorithmic"] * rec["roofline"]["algorithmic_bytes_per_launch"])
                rec["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over "
                                                      f"{pm['launches']} launches of a 1-step sub-run at the same batch; FETCH_SIZE x2 (gfx950 correction of "
                                                      "MI355X_MICROARCH.md), KiB units; fetch "
                                                      f"{pm['fetch_bytes_per_launch']} + write {pm['write_bytes_per_launch']} B per launch there = "
                                                      f"{pm['traffic_over_algorithmic']} x the algorithmic bytes")
                rec["roofline"]["traffic_over_algorithmic"] = pm["traffic_over_algorithmic"]
            else:
                rec["roofline"]["traffic_source"] = (rec["roofline"].get("traffic_source") or "") + f" (live PMC passes failed: {why})"
            rec["pmc_s"] = round(time.perf_counter() - t_pmc, 1)
        if world == 1 and not args.no_cpu_baseline:
            threads = int(os.environ.get("SEPR_CPU_THREADS", str(physical_cores())))
            try:
                rec["cpu_baseline"] = cpu_baseline(VARIANTS[variant], threads)
                rec["speedup_vs_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
            except Exception as e:              # noqa: BLE001 - the line is still worth printing without it
                rec["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        failed = gate_failures(rec)
        if failed:
            rec["gate_failures"] = failed
        rec["summary"] = make_summary(rec)      # last key on purpose (see make_summary)
        emit(rec)
    else:
        failed = []
    sdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if failed:
        print("bench.py: PARITY GATE FAILED: " + "; ".join(failed), file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
