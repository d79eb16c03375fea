"""Lightning-free reader for the reference's `.ckpt` files (test_point_ddpm.py:153-163).

A Lightning checkpoint is a `torch.save`d dict with `state_dict` and `hyper_parameters`
(SURVEY.md section 5).  `hyper_parameters` may be pickled as Lightning's `AttributeDict`;
when Lightning is absent that class is mapped to a plain dict subclass while unpickling.
"""
from __future__ import annotations

import pickle
from typing import Dict, Tuple

import torch


class _AttributeDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "AttributeDict" and ("lightning" in module):
            return _AttributeDict
        return super().find_class(module, name)


class _PickleModule:
    Unpickler = _Unpickler
    load = staticmethod(pickle.load)
    __name__ = "pickle"


def load_lightning_checkpoint(path: str, map_location="cpu") -> Tuple[Dict, Dict[str, torch.Tensor]]:
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    except (ModuleNotFoundError, AttributeError):
        ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)
    if "state_dict" not in ckpt:
        raise RuntimeError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    return dict(ckpt.get("hyper_parameters", {})), ckpt["state_dict"]
