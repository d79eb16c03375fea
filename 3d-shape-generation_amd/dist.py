"""Multi-GPU layer: one process per GPU, batch-sharded sampling, RCCL all-gather of results.

The reference is single-device; every shape is independent through the whole
sampling loop (eval-mode BN, per-shape max-pool/attention, per-sample metrics), so ranks share
nothing per step.  The only collective is one all-gather after the loop: fixed-size per-sample
metric rows, and optionally the output clouds (ragged clouds travel padded + counts).
`torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used for CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if (world > 1 or force_collective()) and not dist.is_initialized():
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def force_collective() -> bool:
    """PCD_DIST_FORCE_COLLECTIVE=1: a world of ONE rank still initialises its process group and runs every gather through
    the backend (no `world == 1` shortcut).  This is how the production branch -- RCCL on device tensors -- executes on
    the single GPU a test box has (`tests/test_gpu_dist.py::test_rccl_branch_on_one_gpu`); it moves no data between
    devices and says nothing about scaling."""
    return os.environ.get("PCD_DIST_FORCE_COLLECTIVE") == "1"


def _no_collective(ws: int) -> bool:
    return ws == 1 and not (force_collective() and dist.is_available() and dist.is_initialized())


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` samples for `rank`; remainders go to the first ranks."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged() -> bool:
    """gloo moves host memory: device tensors are staged through the host (CPU-side tests, or the
    one-GPU rehearsal of the multi-rank path); under nccl (RCCL) device buffers go over xGMI directly."""
    return dist.get_backend() != "nccl"


def all_gather_rows(rows: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Concatenate per-rank row blocks (r_i, ...) in rank order.  Uneven blocks are padded to the
    largest block for one `all_gather_into_tensor` and trimmed afterwards."""
    rank, ws = world()
    if _no_collective(ws):
        return rows
    dev = rows.device
    staged = _host_staged() and dev.type != "cpu"
    if counts is None:
        c = torch.tensor([rows.shape[0]], device="cpu" if staged else dev, dtype=torch.int64)
        cl = [torch.zeros_like(c) for _ in range(ws)]
        dist.all_gather(cl, c)
        counts = [int(v.item()) for v in cl]
    mx = max(counts)
    work = rows.cpu() if staged else rows
    pad = torch.zeros((mx,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    pad[:work.shape[0]] = work
    out = torch.empty((ws * mx,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    out = torch.cat([out[r * mx:r * mx + counts[r]] for r in range(ws)], dim=0)
    return out.to(dev) if staged else out


def all_gather_clouds(clouds: List[torch.Tensor]) -> List[torch.Tensor]:
    """All-gather a ragged python list of (n_i, 3) clouds (LatentDiffusion outputs): sizes first,
    then one padded payload; returns the global list in rank order."""
    rank, ws = world()
    if _no_collective(ws):
        return clouds
    dev = clouds[0].device if clouds else torch.device("cpu")
    sizes = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64, device=dev)
    all_sizes = all_gather_rows(sizes.reshape(-1, 1)).reshape(-1)
    nmax = int(all_sizes.max().item()) if all_sizes.numel() else 0
    pad = torch.zeros(len(clouds), max(nmax, 1), 3, dtype=torch.float32, device=dev)
    for i, c in enumerate(clouds):
        pad[i, :c.shape[0]] = c
    allp = all_gather_rows(pad)
    return [allp[i, :int(all_sizes[i])].clone() for i in range(allp.shape[0])]


class shard_context:
    """While active, `model` draws its on-device noise as samples [lo, lo + n) of a global batch of `total`
    (diffusion._philox_span): ranks get disjoint Philox counter blocks instead of world copies of one stream."""

    def __init__(self, model, lo: int, total: int):
        self.model, self.shard = model, (lo, total)

    def __enter__(self):
        self.prev = getattr(self.model, "_shard", None)
        self.model._shard = self.shard
        return self.model

    def __exit__(self, *exc):
        self.model._shard = self.prev
        return False


def sample_sharded(model, global_batch: int, num_points: int, num_steps: int, x_T_global: Optional[torch.Tensor] = None,
                   sampler: str = "sample", gather: bool = True) -> torch.Tensor:
    """Each rank denoises its shard of the global batch (zero per-step traffic); the clouds are
    all-gathered at the end when `gather`.  With `x_T_global` (host tensor, same on all ranks) the
    result is independent of the number of ranks."""
    rank, ws = world()
    lo, hi = shard_range(global_batch, rank, ws)
    xs = None if x_T_global is None else x_T_global[lo:hi].to(model.device)
    with shard_context(model, lo, global_batch):
        out = getattr(model, sampler)(hi - lo, num_points, num_steps=num_steps, x_T=xs)
    return all_gather_rows(out.contiguous()) if gather else out


def evaluate_sharded(original, reconstructed, use_approximate_gpu_emd: bool = False):
    """Per-sample (CD, EMD, voxel BCE) rows for this rank's samples (test_point_ddpm.py:85-92),
    all-gathered; returns (rows (B_global, 3), their nan-mean).  `original` / `reconstructed` are (B, N, 3)
    tensors or python lists of ragged (n_i, 3) clouds (the latent samplers' output); a pair with an empty
    cloud has no metrics (the reference would raise) and gets a NaN row."""
    from . import metrics
    if len(reconstructed) == 0:
        # an empty local shard (global batch < world size) still takes part in the collective: under nccl its (0, 3)
        # block must live on this rank's device like everybody else's rows, or the all-gather raises / hangs the peers
        on_gpu = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        rows = torch.zeros(0, 3, dtype=torch.float32, device=dev)
    else:
        rows = metrics.pair_metrics(original, reconstructed, use_approximate_gpu_emd)   # one batched enqueue, no host sync per pair
    allrows = all_gather_rows(rows)
    return allrows, torch.nanmean(allrows, dim=0)
