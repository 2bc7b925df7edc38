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
