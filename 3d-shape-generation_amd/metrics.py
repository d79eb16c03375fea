"""Evaluation metrics with the reference's signatures (reference metrics.py), on HIP kernels.

`chamfer_distance`, `normalize_to_cube`, `compute_metrics` run on the device; the exact
Hungarian EMD stays a host-side scipy solve exactly as in the reference (metrics.py:49-92).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .utils import voxelize


def _prep(x: torch.Tensor) -> torch.Tensor:
    if x.device.type != "cuda":
        raise RuntimeError("metrics run only on an MI355X device: move the clouds to 'cuda' (no CPU path)")
    x = x.unsqueeze(0) if x.dim() == 2 else x
    return x.to(torch.float32).contiguous()


def normalize_to_cube(points: torch.Tensor) -> torch.Tensor:
    """metrics.py:7-21: centre on the bbox midpoint, scale by the max |coord| (per cloud)."""
    p = _prep(points)
    out = torch.empty_like(p)
    _lib.check(_lib.load().pcd_normalize_to_cube(p.data_ptr(), p.shape[0], p.shape[1], out.data_ptr(),
                                                 _lib.stream_ptr()), "normalize_to_cube")
    return out


def chamfer_distance(x, y, scaling_factor=1e+3):
    """metrics.py:23-47: mean_i min_j |x_i-y_j| + mean_j min_i |x_i-y_j| (unsquared L2), one scalar
    for the whole batch, times `scaling_factor`.  Distances are direct differences in fp32, which is
    closer to the exact value than the reference's matmul-form `torch.cdist`."""
    xn, yn = normalize_to_cube(x), normalize_to_cube(y)
    if xn.shape[0] != yn.shape[0]:
        raise ValueError("batch sizes must match")
    b, n1, n2 = xn.shape[0], xn.shape[1], yn.shape[1]
    sums = torch.empty(b, 2, dtype=torch.float32, device=xn.device)
    _lib.check(_lib.load().pcd_chamfer_sums(xn.data_ptr(), yn.data_ptr(), b, n1, n2, sums.data_ptr(),
                                            _lib.stream_ptr()), "chamfer_sums")
    tot = sums.sum(dim=0)
    return (tot[0] / (b * n1) + tot[1] / (b * n2)) * scaling_factor


def chamfer_per_sample(x, y, scaling_factor=1e+3) -> torch.Tensor:
    """Chamfer distance of each (x_b, y_b) pair, (B,) -- what test_point_ddpm.py:85-86 loops over."""
    xn, yn = normalize_to_cube(x), normalize_to_cube(y)
    b, n1, n2 = xn.shape[0], xn.shape[1], yn.shape[1]
    sums = torch.empty(b, 2, dtype=torch.float32, device=xn.device)
    _lib.check(_lib.load().pcd_chamfer_sums(xn.data_ptr(), yn.data_ptr(), b, n1, n2, sums.data_ptr(),
                                            _lib.stream_ptr()), "chamfer_sums")
    return (sums[:, 0] / n1 + sums[:, 1] / n2) * scaling_factor


def earth_mover_distance_cpu(x, y, scaling_factor=1):
    """metrics.py:49-92: exact assignment on the host (scipy Hungarian), bug-for-bug divisor
    (`x_pc.shape[1]` is the coordinate dimension 3, not the point count: SURVEY.md A.5)."""
    from scipy.optimize import linear_sum_assignment
    xn, yn = normalize_to_cube(x).cpu().numpy(), normalize_to_cube(y).cpu().numpy()
    if xn.shape[0] != yn.shape[0]:
        raise AssertionError("Batch sizes must be the same")
    vals = []
    for a, b in zip(xn, yn):
        dist = np.linalg.norm(a[:, None] - b[None, :], axis=-1)
        r, c = linear_sum_assignment(dist)
        vals.append(dist[r, c].sum() / max(a.shape[1], b.shape[1]))
    return torch.tensor(vals, device=x.device).mean() * scaling_factor


def earth_mover_distance_gpu(x, y, epsilon=1e-2, thresh=1e-5, max_iter=100, scaling_factor=1):
    """metrics.py:94-158: log-domain Sinkhorn on the device."""
    from .sinkhorn import sinkhorn_emd
    return sinkhorn_emd(normalize_to_cube(x), normalize_to_cube(y), epsilon, thresh, max_iter) * scaling_factor


def voxel_bce(gen: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """BCE between two binary occupancy grids (metrics.py:181): every mismatching voxel costs
    100 (torch clamps log at -100), matching voxels cost 0, mean over all voxels."""
    if gen.shape != ref.shape:
        raise ValueError("grids must have the same shape")
    out = torch.empty(1, dtype=torch.float32, device=gen.device)
    _lib.check(_lib.load().pcd_binary_bce_mean(gen.data_ptr(), ref.data_ptr(), gen.numel(), out.data_ptr(),
                                               _lib.stream_ptr()), "binary_bce_mean")
    return out[0]


def compute_metrics(generated_samples, reference_samples, use_approximate_gpu_emd=False):
    """metrics.py:160-183 -> (chamfer x1e3, EMD, voxel BCE)."""
    avg_cd = chamfer_distance(generated_samples, reference_samples)
    if use_approximate_gpu_emd:
        avg_emd = earth_mover_distance_gpu(generated_samples, reference_samples)
    else:
        avg_emd = earth_mover_distance_cpu(generated_samples, reference_samples)
    recon_loss = voxel_bce(voxelize(generated_samples), voxelize(reference_samples))
    return avg_cd, avg_emd, recon_loss
