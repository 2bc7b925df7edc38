import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The oracle's CPU forward / autograd decides the device suite's wall time (458 - 736 s over the boxes of round 6 for the same code).  On many-core hosts
    torch defaults to one thread per core, which is several times SLOWER than 16 threads for these small ops (bench.py's cpu_baseline sweep on the 128-core
    GPU hosts: 8 threads 0.91 s, 16: 0.73 s, 32: 1.02 s per 4 s forward; all cores ~6x slower).  SEPR_TEST_THREADS overrides; small hosts are left alone."""
    want = os.environ.get("SEPR_TEST_THREADS")
    if want:
        torch.set_num_threads(max(1, int(want)))
    elif (os.cpu_count() or 1) > 32 and torch.get_num_threads() > 16:
        torch.set_num_threads(16)
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


# A/B switches the library latches once per process (include/sepr.h SEPR_KNOB_*): a test that flips one with monkeypatch.setenv gets the
# library's copy refreshed right away, and restored together with the environment when the test ends.
_KNOBS = {"SEPR_X3_WIDE", "SEPR_TRAIN_GCFN_PLANES", "SEPR_TRAIN_ATTN_ONE", "SEPR_TRAIN_CLA16", "SEPR_FOLD_HEAD", "SEPR_TN16"}


def _knobs_reload():
    from sepreformer_amd import lib as L
    if os.path.exists(L.LIB_PATH):
        L.load().sepr_knobs_reload()


@pytest.fixture(autouse=True)
def _knob_env(monkeypatch):
    touched = []
    plain_setenv = monkeypatch.setenv

    def setenv(name, value, prepend=None):
        plain_setenv(name, value, prepend)
        if name in _KNOBS:
            touched.append(name)
            _knobs_reload()

    monkeypatch.setenv = setenv
    yield
    if touched:
        monkeypatch.undo()
        _knobs_reload()
