"""Voxel <-> point conversions of the hot path (reference utils.py:488-539) on HIP kernels,
plus the small host helpers the entry points need (logger)."""
from __future__ import annotations

import logging
import os
from typing import List

import torch

from . import _lib


def voxelize(points: torch.Tensor, voxel_resolution: int = 32) -> torch.Tensor:
    """utils.py:488-509: (B,N,3) in [-1,1] -> occupancy (B,R,R,R) indexed [x][y][z]."""
    if points.device.type != "cuda":
        raise RuntimeError("voxelize runs only on an MI355X device (no CPU path)")
    p = points.unsqueeze(0) if points.dim() == 2 else points
    p = p.to(torch.float32).contiguous()
    b, n, r = p.shape[0], p.shape[1], voxel_resolution
    vox = torch.empty(b, r, r, r, dtype=torch.float32, device=p.device)
    lib = _lib.load()
    _lib.check(lib.pcd_fill_zero(vox.data_ptr(), vox.numel() * 4, _lib.stream_ptr()), "fill_zero")
    _lib.check(lib.pcd_voxelize(p.data_ptr(), b, n, r, vox.data_ptr(), _lib.stream_ptr()), "voxelize")
    return vox


def voxel_tensor_to_point_clouds(voxel_grid: torch.Tensor, threshold: float = 0.5) -> List[torch.Tensor]:
    """utils.py:511-539: (B,1,D,H,W) -> python list of (n_i,3) clouds in [-1,1], points ordered by
    the row-major (z,y,x) scan of `torch.where`, columns [x,y,z].  Empty grids give (0,3)."""
    if voxel_grid.device.type != "cuda":
        raise RuntimeError("voxel_tensor_to_point_clouds runs only on an MI355X device (no CPU path)")
    b, _, d, h, w = voxel_grid.shape
    v = voxel_grid.to(torch.float32).contiguous()
    counts = torch.empty(b, dtype=torch.int32, device=v.device)
    pts = torch.empty(b, d * h * w, 3, dtype=torch.float32, device=v.device)
    _lib.check(_lib.load().pcd_voxels_to_points(v.data_ptr(), b, d, h, w, float(threshold), counts.data_ptr(),
                                                pts.data_ptr(), _lib.stream_ptr()), "voxels_to_points")
    cnt = counts.cpu().tolist()          # the ragged python list forces one host sync, as in the reference
    return [pts[i, :cnt[i]].clone() for i in range(b)]


def setup_logger(name: str, log_file: str, level=logging.INFO) -> logging.Logger:
    """File + console logger (reference utils.py:354-385 behaviour)."""
    os.makedirs(os.path.dirname(log_file) or ".", exist_ok=True)
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
        for h in (logging.FileHandler(log_file), logging.StreamHandler()):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger
