"""bench.py end to end on the device (round-2 review item 3c): the N > 1 code path with two ranks sharing GPU 0 (debug mode
``--share-gpu``: gloo collectives, the same sharding / reduction / timing code as the RCCL run the driver launches on a
multi-GPU node), for the inference line and for the training line, and the one-rank RCCL group of the default 1-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_share_gpu_inference():
    rec = _bench("--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-precision")
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["collective_backend"] == "gloo"
    assert rec["reduced_metric"]["utterances"] == 64               # 32 per rank, summed by the all-reduce
    assert rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["parity_db_vs_golden"] >= 80.0 and rec["pit_si_snr_max_abs_delta_db"] <= 1e-3


def test_bench_two_ranks_share_gpu_training():
    rec = _bench("--gpus", "2", "--share-gpu", "--mode", "train", "--batch", "2", "--steps", "1", "--warmup", "1")
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2
    assert rec["allreduce_bytes_per_step"] > 50_000_000            # the whole flat gradient buffer (58.8 MB for Base), two buckets
    assert rec["loss"] == rec["loss"] and rec["grad_norm"] > 0      # finite


def test_bench_single_gpu_runs_its_collectives_through_rccl():
    """World size 1: bench.py still creates a (one-rank) nccl process group, so the metric reduction of every step and the
    training gradient all-reduce execute RCCL on this box too."""
    rec = _bench("--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-precision", "--batch", "4")
    assert rec["rccl_ranks"] == 1 and rec["collective_backend"] == "nccl", rec.get("collective_error")
    assert rec["reduced_metric"]["utterances"] == 4
    rec = _bench("--mode", "train", "--batch", "2", "--steps", "1", "--warmup", "1")
    assert rec["collective_backend"] == "nccl" and rec["allreduce_bytes_per_step"] > 50_000_000


def test_bench_measures_hbm_traffic_live():
    """--pmc on: the roofline's ``traffic`` comes from rocprofv3 FETCH_SIZE / WRITE_SIZE passes of THIS invocation (separate passes,
    kernel-trace only), not from a committed file; the fused GCFN kernel moves more than its algorithmic bytes and less than 3 x."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    rec = _bench("--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-precision", "--pmc", "on", timeout=600)
    roof = rec["roofline"]
    assert roof["traffic_source"].startswith("measured in this run"), roof["traffic_source"]
    assert 1.0 < roof["traffic_over_algorithmic"] < 3.0, roof
    assert roof["traffic"] == round(roof["traffic_over_algorithmic"] * roof["algorithmic_bytes_per_launch"])


# ---------------------------------------------------------------------------------------------------------------------
# data-parallel gradients: two ranks (sharing GPU 0, gloo) against the single-rank computation
# ---------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import dataclasses
    import numpy as np
    import torch
    from sepreformer_amd import dist as sd
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    from sepreformer_amd.synth import synth_sources
    sd.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dataclasses.replace(VARIANTS["tiny"], dropout=0.0)
    src = torch.from_numpy(synth_sources(2, 1500, seed=50 + rank) * 4.0).to(dev)
    x = src.sum(1).contiguous()

    def step(sync):
        m = Model.from_config(cfg, init_seed=0).load_synthetic_(0).to(dev).train()
        m.grad_sync = sync
        audio, aux = m(x)
        loss = sum(((a - src[:, s, : a.shape[-1]]) ** 2).mean() for s, a in enumerate(audio)) + 0.1 * sum(torch.stack(a_).abs().mean() for a_ in aux)
        loss.backward()
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
        bn = torch.cat([v.reshape(-1).float() for k, v in m.state_dict().items() if "running_" in k])
        return flat, bn

    g_local, bn_local = step(None)
    sync = sd.GradSync()
    g_sync, bn_sync = step(sync)
    assert sync.calls == 1 and sync.bytes == g_sync.numel() * 4
    np.savez(os.path.join(out_dir, f"dp{rank}.npz"), local=g_local.cpu().numpy(), synced=g_sync.cpu().numpy(),
             bn_local=bn_local.cpu().numpy(), bn_sync=bn_sync.cpu().numpy())
    sd.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradients_are_the_mean_of_the_single_rank_gradients(tmp_path):
    """configs[4]'s data-parallel step, checked without a multi-GPU node: two ranks (sharing GPU 0, gloo - the same GradSync code path
    the RCCL run takes) each back-propagate their own batch through the HIP training path; the synchronised gradient on BOTH ranks must
    be exactly (g_0 + g_1) / 2 of the two single-rank gradients (fp32 sum of two terms: order-independent, so the check is bitwise),
    i.e. the gradient of the batch-mean loss over the union of the shards with per-rank BatchNorm statistics - the reference's
    per-replica statistics (engine.py:64).  BatchNorm running statistics stay per rank: untouched by the synchronisation."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "dp0.npz"), np.load(tmp_path / "dp1.npz")
    want = (r0["local"] + r1["local"]) / np.float32(2.0)
    assert np.abs(r0["local"] - r1["local"]).max() > 0             # the shards differ
    assert np.array_equal(r0["synced"], want) and np.array_equal(r1["synced"], want)
    assert np.array_equal(r0["bn_sync"], r0["bn_local"]) and np.array_equal(r1["bn_sync"], r1["bn_local"])
    assert np.abs(r0["bn_local"] - r1["bn_local"]).max() > 0       # per-rank statistics
