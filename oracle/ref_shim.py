"""Import the read-only reference (`/root/reference`) in THIS container.

TEST TOOLING, container-only: the reference never travels to the GPU box, so
nothing under `tests -m gpu`, `smoke()` or `bench.py` may call this.  It is
used by `oracle/make_golden.py` (fixture capture) only.

The reference needs two modules that are not installed and carry no
arithmetic: `pytorch_lightning` (base class + hparams capture) and `plyfile`
(PLY writer).  They are replaced by the minimal stand-ins below, following the
recipe recorded in SURVEY.md section 8(c).  `load_reference_data()` additionally
stands in for `deepdish` (HDF5 file reader, no arithmetic either): `dd.io.load`
returns the `data` array of an `.npz` written under the `.dd` name.
"""
from __future__ import annotations

import inspect
import os
import sys
import types

REFERENCE_DIR = os.environ.get("PCD_REFERENCE_DIR", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "diffusion.py"))


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _install_stubs():
    import torch
    import torch.nn as nn

    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            def __init__(self):
                super().__init__()
                self.logger = None
                self.trainer = None
                self.current_epoch = 0

            def save_hyperparameters(self, ignore=None):
                frame = inspect.currentframe().f_back
                info = inspect.getargvalues(frame)
                hp = _AttrDict()
                for a in info.args[1:]:
                    if ignore and a in ignore:
                        continue
                    hp[a] = info.locals[a]
                object.__setattr__(self, "_hparams_store", hp)

            @property
            def hparams(self):
                return self._hparams_store

            @property
            def device(self):
                return next(self.parameters()).device

            def log(self, *a, **k):
                pass

        pl.LightningModule = LightningModule
        pl.LightningDataModule = object
        pl.seed_everything = lambda s: torch.manual_seed(s)
        sys.modules["pytorch_lightning"] = pl
    if "plyfile" not in sys.modules:
        pf = types.ModuleType("plyfile")
        pf.PlyData = object
        pf.PlyElement = object
        sys.modules["plyfile"] = pf


def load_reference():
    """Returns the reference's (diffusion, networks, metrics, utils) modules."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_DIR}")
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    _install_stubs()
    # The reference is a flat directory of modules named diffusion/networks/metrics/utils.
    saved = {k: sys.modules.pop(k) for k in ("diffusion", "networks", "metrics", "utils") if k in sys.modules}
    sys.path.insert(0, REFERENCE_DIR)
    try:
        import diffusion as r_diffusion
        import networks as r_networks
        import metrics as r_metrics
        import utils as r_utils
    finally:
        sys.path.remove(REFERENCE_DIR)
        for k in ("diffusion", "networks", "metrics", "utils"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    return r_diffusion, r_networks, r_metrics, r_utils


def load_reference_data():
    """The reference's data.py module, with `deepdish.io.load(path)` reading {'data': array} from an npz file."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_DIR}")
    import numpy as np
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    _install_stubs()
    if "deepdish" not in sys.modules:
        dd = types.ModuleType("deepdish")
        dd.io = types.ModuleType("deepdish.io")

        def _load(path):
            with np.load(path) as f:
                return {"data": f["data"]}

        dd.io.load = _load
        sys.modules["deepdish"] = dd
        sys.modules["deepdish.io"] = dd.io
    saved = sys.modules.pop("data", None)
    sys.path.insert(0, REFERENCE_DIR)
    try:
        import data as r_data
    finally:
        sys.path.remove(REFERENCE_DIR)
        sys.modules.pop("data", None)
        if saved is not None:
            sys.modules["data"] = saved
    return r_data
