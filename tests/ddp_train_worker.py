"""Worker for tests/test_gpu_train_ddp.py: two ranks (both on cuda:0, gloo rendezvous on 127.0.0.1) run two data-parallel
training steps of the point denoiser on different batches; rank 0 also replays the same two steps single-process on the
mean of the gathered gradients (AdamW's first step in closed form) to check the arithmetic of the exchange."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shapegen_amd  # noqa: E402,F401
from helpers import point_sd  # noqa: E402
from shapegen_amd.diffusion import PointCloudDiffusion  # noqa: E402
from shapegen_amd.training import PointTrainer  # noqa: E402


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(2, 128, 3, generator=g) * 0.5, torch.rand(2, generator=g), torch.randn(2, 128, 3, generator=g)


def main():
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=int(os.environ["RANK"]),
                            world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    model = PointCloudDiffusion(num_points=128)
    model.load_state_dict(point_sd(), strict=True)
    model = model.to("cuda")
    tr = PointTrainer(model.model, lr=1e-3)
    local_grads, after_first = [], None
    p_start = tr.P.detach().cpu().double()
    for step in range(2):
        x, t, n = batch(10 * step + rank)
        tr.forward(x.cuda(), t.cuda())
        tr.backward(n.cuda())
        local_grads.append(tr.G.clone())
        tr.optimizer_step()
        if step == 0:
            after_first = tr.P.detach().cpu().double()
    flat = tr.P.detach().cpu()
    # every rank holds the same parameters after the exchange
    ref = flat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(flat, ref), "ranks diverged"
    # the first update used the MEAN of the two ranks' gradients: replay AdamW step 1 on the host from the gathered gradients
    g0 = [torch.zeros_like(local_grads[0].cpu()) for _ in range(world)]
    dist.all_gather(g0, local_grads[0].cpu())
    if rank == 0:
        mean_g = (sum(g.double() for g in g0) / world / tr.loss_scale)
        # AdamW step 1 in closed form: m_hat / (sqrt(v_hat) + eps) = g / (|g| + eps)
        want = p_start * (1 - 1e-3 * 1e-5) - 1e-3 * mean_g / (mean_g.abs() + 1e-8)
        np.save(os.environ["DDP_OUT"], np.array([float((g0[0] - g0[1]).abs().max()), float((after_first - want).abs().max())]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
