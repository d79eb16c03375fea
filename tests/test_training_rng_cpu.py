"""Data-parallel `training.fit`: ranks must enumerate ONE shuffled batch sequence (consecutive batches are dealt out to
the ranks) while drawing different (t, noise, dropout) streams.  `_RankRng` keeps the global torch RNG shared and swaps a
per-rank state in around the model's own draws (a per-rank `torch.manual_seed` before the epoch loop gave every
rank its own permutation, so an epoch no longer partitioned the dataset)."""
import torch
from torch.utils.data import DataLoader, TensorDataset

import shapegen_amd  # noqa: F401
from shapegen_amd.training import _RankRng


def _epoch(rank):
    """What one rank of fit() does with the RNG: iterate a shuffled loader, draw under its private state per batch."""
    torch.manual_seed(1234)
    rr = _RankRng("cpu", rank)
    order, draws = [], []
    for (b,) in DataLoader(TensorDataset(torch.arange(40)), batch_size=4, shuffle=True):
        order.append(b.tolist())
        with rr:
            draws.append((torch.rand(2).tolist(), torch.initial_seed()))
    return order, draws, torch.initial_seed(), torch.rand(3)


def test_ranks_share_the_shuffle_and_not_the_noise():
    o0, d0, seed0, after0 = _epoch(0)
    o1, d1, seed1, after1 = _epoch(1)
    assert o0 == o1                                             # the same permutation on every rank
    assert sorted(sum(o0, [])) == list(range(40))               # which is a partition of the dataset
    assert all(a[0] != b[0] for a, b in zip(d0, d1))            # different t / noise draws for every batch
    assert d0[0][1] != d1[0][1] and d0[0][1] != 1234            # the Philox seed (torch.initial_seed()) is per rank inside
    assert seed0 == seed1 == 1234                               # and the process-wide seed is back outside: nothing leaks
    assert torch.equal(after0, after1)
    assert len({tuple(d[0]) for d in d0}) == len(d0)            # a rank's stream advances from batch to batch
