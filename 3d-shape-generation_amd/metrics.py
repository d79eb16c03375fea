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
    closer to the exact value than the reference's matmul-form `torch.cdist`.
    Runs through the batched pair kernels (`pcd_pair_metrics`: queries x target splits x pairs blocks, so a single
    cloud pair still spreads over the chip); with equal sizes the batch-joint mean is the mean of the per-pair values."""
    return chamfer_per_sample(x, y, scaling_factor).mean()


def chamfer_per_sample(x, y, scaling_factor=1e+3) -> torch.Tensor:
    """Chamfer distance of each (x_b, y_b) pair, (B,) -- what test_point_ddpm.py:85-86 loops over."""
    x, y = _prep(x), _prep(y)
    if x.shape[0] != y.shape[0]:
        raise ValueError("batch sizes must match")
    return _pair_rows(x, y, False)[:, 0] * scaling_factor


def _pair_rows(a: torch.Tensor, b: torch.Tensor, sinkhorn: bool, epsilon=1e-2, thresh=1e-5, max_iter=100) -> torch.Tensor:
    """pcd_pair_metrics on dense (P, N, 3) / (P, M, 3) clouds: rows (P, 3) = (Chamfer x1, Sinkhorn EMD | 0, voxel BCE)."""
    P, n, m = a.shape[0], a.shape[1], b.shape[1]
    dev = a.device
    na = torch.full((P,), n, dtype=torch.int32, device=dev)
    nb = torch.full((P,), m, dtype=torch.int32, device=dev)
    log_mu = torch.log(torch.ones(P) / n + 1e-10).to(dev)
    log_nu = torch.log(torch.ones(P) / m + 1e-10).to(dev)
    lib = _lib.load()
    out = torch.empty(P, 3, dtype=torch.float32, device=dev)
    need = int(lib.pcd_pair_metrics_workspace_bytes(P, n, m))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.pcd_pair_metrics(a.data_ptr(), na.data_ptr(), n, b.data_ptr(), nb.data_ptr(), m, P, 1 if sinkhorn else 0,
                                    float(epsilon), float(thresh), int(max_iter), log_mu.data_ptr(), log_nu.data_ptr(),
                                    out.data_ptr(), ws.data_ptr(), need, _lib.stream_ptr()), "pair_metrics")
    return out


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
    x, y = _prep(x), _prep(y)
    if x.shape[0] == 1 and y.shape[0] == 1:
        # one pair: batch-global C.max() and convergence test ARE per pair, so the device-resident loop applies
        return _pair_rows(x, y, True, epsilon, thresh, max_iter)[0, 1] * scaling_factor
    from .sinkhorn import sinkhorn_emd           # batch-joint cost normalisation and convergence test, as the reference
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


def pair_metrics(a_clouds, b_clouds, use_approximate_gpu_emd=False, epsilon=1e-2, thresh=1e-5, max_iter=100) -> torch.Tensor:
    """(P, 3) rows [Chamfer x1e3, EMD, voxel BCE] of the independent pairs (a_p, b_p): row p is what
    `compute_metrics(a_p, b_p, use_approximate_gpu_emd)` returns for that pair alone (the per-sample loop of
    test_point_ddpm.py:85-92), but all pairs run in ONE enqueue of batched kernels (`pcd_pair_metrics`) with no host
    synchronisation.  `a_clouds` / `b_clouds`: (P, N, 3) tensors or python lists of ragged (n_i, 3) clouds; a pair with
    an empty cloud has no metrics (the reference would raise) and gets a NaN row.  The exact Hungarian EMD
    (use_approximate_gpu_emd=False) stays a host-side scipy solve per pair, exactly as in the reference."""
    P = len(a_clouds)
    dev = a_clouds[0].device if P else torch.device("cuda")
    rows = torch.full((P, 3), float("nan"), dtype=torch.float32, device=dev)
    keep = [i for i in range(P) if a_clouds[i].shape[0] > 0 and b_clouds[i].shape[0] > 0]
    if not keep:
        return rows
    if dev.type != "cuda":
        raise RuntimeError("metrics run only on an MI355X device: move the clouds to 'cuda' (no CPU path)")

    def pack(clouds):
        counts = [int(clouds[i].shape[0]) for i in keep]
        nmax = max(counts)
        if isinstance(clouds, torch.Tensor) and len(keep) == P:
            return clouds.to(torch.float32).contiguous(), counts, nmax
        buf = torch.zeros(len(keep), nmax, 3, dtype=torch.float32, device=dev)
        for j, i in enumerate(keep):
            buf[j, :counts[j]] = clouds[i]
        return buf, counts, nmax

    a, ca, na_max = pack(a_clouds)
    b, cb, nb_max = pack(b_clouds)
    na = torch.tensor(ca, dtype=torch.int32, device=dev)
    nb = torch.tensor(cb, dtype=torch.int32, device=dev)
    # log(mu + 1e-10) with the reference's fp32 torch ops (metrics.py:133-134,141), one value per pair
    log_mu = torch.log(torch.ones(len(keep)) / torch.tensor(ca, dtype=torch.float32) + 1e-10).to(dev)
    log_nu = torch.log(torch.ones(len(keep)) / torch.tensor(cb, dtype=torch.float32) + 1e-10).to(dev)
    lib = _lib.load()
    out = torch.empty(len(keep), 3, dtype=torch.float32, device=dev)
    need = int(lib.pcd_pair_metrics_workspace_bytes(len(keep), na_max, nb_max))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.pcd_pair_metrics(a.data_ptr(), na.data_ptr(), na_max, b.data_ptr(), nb.data_ptr(), nb_max, len(keep),
                                    1 if use_approximate_gpu_emd else 0, float(epsilon), float(thresh), int(max_iter),
                                    log_mu.data_ptr(), log_nu.data_ptr(), out.data_ptr(), ws.data_ptr(), need,
                                    _lib.stream_ptr()), "pair_metrics")
    out[:, 0] *= 1e3                                         # chamfer_distance's default scaling_factor
    if not use_approximate_gpu_emd:
        for j, i in enumerate(keep):
            out[j, 1] = earth_mover_distance_cpu(a_clouds[i], b_clouds[i]).to(torch.float32)
    rows[torch.tensor(keep, device=dev)] = out
    return rows


def compute_metrics(generated_samples, reference_samples, use_approximate_gpu_emd=False):
    """metrics.py:160-183 -> (chamfer x1e3, EMD, voxel BCE)."""
    avg_cd = chamfer_distance(generated_samples, reference_samples)
    if use_approximate_gpu_emd:
        avg_emd = earth_mover_distance_gpu(generated_samples, reference_samples)
    else:
        avg_emd = earth_mover_distance_cpu(generated_samples, reference_samples)
    recon_loss = voxel_bce(voxelize(generated_samples), voxelize(reference_samples))
    return avg_cd, avg_emd, recon_loss
