"""sepreformer_amd - the SepReformer separator forward path for AMD MI355X (gfx950).

Package layout (only what the hot path needs):

    csrc/      hand-written HIP kernels + the extern "C" boundary (include/sepr.h)
    lib.py     ctypes binding of libsepr_hip.so (raises if the library is not built)
    pack.py    host-side weight packing (state_dict -> kernel layouts, BatchNorm folding)
    engine.py  forward driver: separator topology -> one C-ABI call per fused block
    model.py   ``Model``: the reference's nn.Module / state_dict / configs.yaml surface
    params.py  declarative parameter tree with the reference's state_dict names
    config.py  hyper-parameters, named variants
    synth.py   deterministic synthetic weights / mixtures (the real checkpoint is not available)
    dist.py    utterance sharding across ranks (one process per GPU, RCCL)
"""
from .config import SepConfig, VARIANTS, load_model_kwargs  # noqa: F401

__all__ = ["SepConfig", "VARIANTS", "load_model_kwargs", "Model"]


def __getattr__(name):
    if name == "Model":
        from .model import Model
        return Model
    raise AttributeError(name)
