"""Worker for tests/test_gpu_dist.py::test_rccl_branch_on_one_gpu: ONE rank, backend "nccl" (= RCCL), device_id = cuda:0,
PCD_DIST_FORCE_COLLECTIVE=1 so that no `world == 1` shortcut is taken: every gather of `shapegen_amd.dist` and every
collective of `bench.Ranks` runs through RCCL on device tensors -- the branch an 8-GPU job takes, executed on the one GPU
this box has.  Nothing moves between devices and nothing here is a scaling measurement."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shapegen_amd  # noqa: E402,F401
import bench  # noqa: E402
from shapegen_amd import dist as D  # noqa: E402
from shapegen_amd import metrics as M  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    assert D.force_collective()
    R = bench.Ranks(argparse.Namespace(gpus=1))              # builds the one-rank nccl group with device_id=cuda:0
    import torch.distributed as dist
    res = {"backend": dist.get_backend(), "ranks_backend": R.backend, "world": dist.get_world_size(),
           "host_staged": D._host_staged()}
    dev = torch.device("cuda", 0)
    # fixed-size metric rows, with and without the size exchange
    rows = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
    a = D.all_gather_rows(rows)
    b = D.all_gather_rows(rows, counts=[4])
    res["rows_equal"] = bool(torch.equal(a, rows) and torch.equal(b, rows) and a.device == dev and a.data_ptr() != rows.data_ptr())
    # ragged clouds (the latent samplers' outputs), one of them empty
    g = torch.Generator().manual_seed(3)
    clouds = [torch.randn(n, 3, generator=g).to(dev) for n in (17, 0, 301, 64)]
    got = D.all_gather_clouds(clouds)
    res["clouds_equal"] = bool(len(got) == 4 and all(torch.equal(x, y) for x, y in zip(got, clouds)))
    # sharded evaluation rows through the device all-gather
    orig = [torch.tanh(torch.randn(n, 3, generator=g)).to(dev) for n in (200, 150, 90)]
    recon = [c + 0.01 for c in orig]
    allrows, mean = D.evaluate_sharded(orig, recon, use_approximate_gpu_emd=True)
    want = M.pair_metrics(orig, recon, True)
    res["eval_equal"] = bool(torch.equal(allrows, want) and allrows.device == dev)
    res["mean_ok"] = bool(torch.allclose(mean, want.mean(0)))
    # batch-sharded sampler with a one-rank shard (gather on)
    from helpers import point_sd
    from shapegen_amd.diffusion import PointCloudDiffusion
    model = PointCloudDiffusion(num_points=128)
    model.load_state_dict(point_sd(), strict=True)
    model = model.to(dev).eval()
    xT = torch.randn(3, 128, 3, generator=g)
    sharded = D.sample_sharded(model, 3, 128, 4, x_T_global=xT)
    res["sampler_equal"] = bool(torch.equal(sharded, model.sample(3, 128, num_steps=4, x_T=xT.to(dev))))
    # bench.py's plumbing on the data-path backend
    res["max_over_ranks"] = R.max_over_ranks(1.25)
    res["collective_ranks"] = R.collective_ranks()
    R.sync_all()
    R.finish()
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
