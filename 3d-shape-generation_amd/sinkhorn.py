"""Host loop of the log-domain Sinkhorn EMD (reference metrics.py:94-158) over HIP kernels."""
from __future__ import annotations

import torch

from . import _lib


def sinkhorn_emd(x: torch.Tensor, y: torch.Tensor, epsilon=1e-2, thresh=1e-5, max_iter=100) -> torch.Tensor:
    """x (B,n,3), y (B,m,3) already normalised to the cube.  Returns mean_b sum_ij P_ij C_ij."""
    lib = _lib.load()
    st = _lib.stream_ptr()
    b, n, m = x.shape[0], x.shape[1], y.shape[1]
    dev = x.device
    cmax = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(lib.pcd_pairwise_max_dist(x.data_ptr(), y.data_ptr(), b, n, m, cmax.data_ptr(), st), "pair_max")
    alpha = torch.zeros(b, n, dtype=torch.float32, device=dev)
    beta = torch.zeros(b, m, dtype=torch.float32, device=dev)
    # log(mu + 1e-10) with the reference's fp32 torch ops (metrics.py:133-134,141)
    log_mu = float(torch.log(torch.ones(1) / n + 1e-10))
    log_nu = float(torch.log(torch.ones(1) / m + 1e-10))
    err = torch.empty(2, dtype=torch.float32, device=dev)
    for _ in range(max_iter):
        _lib.check(lib.pcd_sinkhorn_dual_update(x.data_ptr(), y.data_ptr(), b, n, m, cmax.data_ptr(), epsilon, log_mu,
                                                beta.data_ptr(), alpha.data_ptr(), err.data_ptr(), st), "sinkhorn alpha")
        _lib.check(lib.pcd_sinkhorn_dual_update(y.data_ptr(), x.data_ptr(), b, m, n, cmax.data_ptr(), epsilon, log_nu,
                                                alpha.data_ptr(), beta.data_ptr(), err.data_ptr() + 4, st), "sinkhorn beta")
        e = err.cpu()                      # one host sync per iteration, like the reference's `if err < thresh`
        if float(e[0]) < thresh and float(e[1]) < thresh:
            break
    scratch = torch.empty(b, n, dtype=torch.float32, device=dev)
    cost = torch.empty(b, dtype=torch.float32, device=dev)
    _lib.check(lib.pcd_sinkhorn_cost(x.data_ptr(), y.data_ptr(), b, n, m, cmax.data_ptr(), epsilon, alpha.data_ptr(),
                                     beta.data_ptr(), scratch.data_ptr(), cost.data_ptr(), st), "sinkhorn cost")
    return cost.mean()
