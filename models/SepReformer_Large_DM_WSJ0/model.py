"""`models.SepReformer_Large_DM_WSJ0.model.Model` - same import path and call contract as the reference
(`/root/reference/models/SepReformer_Large_DM_WSJ0/model.py:12-54`): `Model(**yaml["config"]["model"])`, `forward(x)`.

In the reference the name is bound to a logging wrapper *function* around the class
(`utils/decorators.py:4-16`); callers only ever call it, so a factory keeps the contract.
"""
from sepreformer_amd.model import Model as _Model


def Model(**kwargs):
    kwargs.setdefault("per_level_split", False)
    return _Model(**kwargs)
