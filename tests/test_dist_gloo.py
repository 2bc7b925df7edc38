"""N>1 path on CPU: two gloo ranks shard a batch by utterance, separate their slice, gather, and reduce a
metric.  The compute function here is the oracle (allowed in tests); the sharding / gather / reduce
logic under test is the product's (sepreformer_amd/dist.py)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import sepreformer_oracle as orc
    from sepreformer_amd import dist as sd
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.synth import synth_mixture, synth_state_dict
    r, w, _ = sd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = VARIANTS["tiny"]
    weights = synth_state_dict(cfg, 0)
    x = synth_mixture(total, 600, seed=99) * 4

    def separate(xb):
        audio, _ = orc.model_forward(weights, cfg, xb)
        return torch.stack([a.reshape(xb.shape[0], -1) for a in audio], 0)

    full, (a, b) = sd.separate_sharded(separate, x, gather=True, chunk=2)
    local, (a2, b2) = sd.separate_sharded(separate, x, gather=False, chunk=2) if b > a else (None, (a, b))
    sums = sd.reduce_metric_sums(torch.tensor([float(b - a), float(full[:, a:b].abs().sum())], dtype=torch.float64))
    t = sd.max_over_ranks(1.0 + rank, torch.device("cpu"))
    sd.barrier()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full.numpy(), rng=np.array([a, b]), sums=sums.numpy(), t=t)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("total", [5, 1])
def test_two_rank_sharded_inference(tmp_path, total):
    sys.path.insert(0, ROOT)
    from oracle import sepreformer_oracle as orc
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.dist import shard_range
    from sepreformer_amd.synth import synth_mixture, synth_state_dict
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    cfg = VARIANTS["tiny"]
    x = synth_mixture(total, 600, seed=99) * 4
    audio, _ = orc.model_forward(synth_state_dict(cfg, 0), cfg, x)
    want = torch.stack([a.reshape(total, -1) for a in audio], 0).numpy()
    got = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in range(world):
        assert tuple(got[r]["rng"]) == shard_range(total, r, world)
        # every rank holds the full gathered result and it equals the unsharded run (utterances are
        # independent in eval mode: SURVEY.md section 8e)
        assert orc.agreement_db(torch.from_numpy(got[r]["full"]), torch.from_numpy(want)) > 100
        assert got[r]["sums"][0] == total
        assert abs(got[r]["sums"][1] - np.abs(want).sum()) < 1e-3 * np.abs(want).sum()
        assert got[r]["t"] == 2.0


def test_shard_range_properties():
    from sepreformer_amd.dist import shard_range
    for total in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (the driver invokes bench.py directly).
    Here there is no GPU, so every rank stops at the "needs an MI355X" check - which proves the ranks were started with
    the torchrun environment (WORLD_SIZE=2) instead of dying at an argument check."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    txt = r.stdout + r.stderr
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ngpu >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["reduced_metric"]["utterances"] == 64
        return
    assert r.returncode != 0
    assert "but WORLD_SIZE" not in txt, txt[-2000:]                 # the ranks were started with the torchrun environment
    if ngpu == 0:
        # BOTH ranks got past the argument / WORLD_SIZE checks and stopped at the device check, each with the exact message
        assert txt.count("bench.py needs an MI355X") == 2, txt[-3000:]
    else:
        # one visible GPU: rank 0 runs, rank 1 has no device (the shared-GPU debug mode is covered by tests/test_bench_gpu.py)
        assert "invalid device ordinal" in txt or "device ordinal" in txt.lower() or "hipErrorInvalidDevice" in txt, txt[-3000:]


def _gradsync_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sepreformer_amd import dist as sd
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.train_pack import GradBuffer
    sd.init_from_env("gloo")
    gb = GradBuffer(VARIANTS["tiny"], torch.device("cpu"))
    g = torch.Generator().manual_seed(100 + rank)
    gb.flat.copy_(torch.randn(gb.numel, generator=g))
    sync = sd.GradSync()
    tail = gb.offsets["separator.simple_fusion.0.weight"][0]
    sync.begin(gb.flat, tail)            # decoder half first (asynchronous), as the training backward does
    sync(gb.flat)                        # then the head, the wait and the 1/world scale
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), w=gb.view("separator.simple_fusion.0.weight").numpy(), calls=sync.calls, nbytes=sync.bytes,
             numel=gb.numel)
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_sync(tmp_path):
    """Training DP (BASELINE configs[4]): the flat gradient buffer of every rank is replaced by the mean over ranks in ONE
    all-reduce; parameter views into the buffer see the averaged values."""
    sys.path.insert(0, ROOT)
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.train_pack import GradBuffer
    world = 2
    mp.spawn(_gradsync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    gb = GradBuffer(VARIANTS["tiny"], torch.device("cpu"))
    flats = [torch.randn(gb.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    gb.flat.copy_(sum(flats) / world)
    want = gb.view("separator.simple_fusion.0.weight").numpy()
    for r in range(world):
        got = np.load(tmp_path / f"g{r}.npz")
        assert np.allclose(got["w"], want, atol=1e-6)
        assert int(got["calls"]) == 1 and int(got["nbytes"]) == 4 * gb.numel
    # layout contract of the flat buffer: parameter order, 256-byte aligned slices, exact shapes
    names = [n for n, _ in Model_param_names()]
    assert list(gb.offsets) == names
    assert all(off % 64 == 0 for off, _ in gb.offsets.values())


def Model_param_names():
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    return list(Model.from_config(VARIANTS["tiny"], init_seed=0).named_parameters())
