"""MI355X-native point-cloud diffusion sampler (hot path of dhillon24/3d-shape-generation).

Import as `shapegen_amd` (see `/shapegen_amd.py`).  Heavy modules are imported lazily so
that `specs` (pure numpy) is usable without torch or the HIP library.
"""
from . import specs  # noqa: F401

__all__ = ["specs"]
