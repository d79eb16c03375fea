"""Entry point kept from the reference (reference train_point_ddpm.py): constructs
`PointCloudDiffusion(num_points=2048)` and runs the post-training `model.sample(10, 2048)` tail
(:93-99) on the HIP sampler.  `trainer.fit` (training forward/backward) is outside this
framework's scope: pass --ckpt to sample from trained weights.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

import shapegen_amd  # noqa: F401
from shapegen_amd import specs
from shapegen_amd.diffusion import PointCloudDiffusion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None, help="reference Lightning .ckpt to load (otherwise synthetic weights)")
    ap.add_argument("--num-points", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join("samples", "point_cloud_diffusion"))
    args = ap.parse_args()
    torch.manual_seed(24)
    if args.ckpt:
        model = PointCloudDiffusion.load_from_checkpoint(args.ckpt)
    else:
        model = PointCloudDiffusion(num_points=args.num_points)
        sd = specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=1.3)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        print("no --ckpt given: training is out of scope here, sampling from synthetic weights")
    model = model.to("cuda").eval()
    samples = model.sample(10, args.num_points, num_steps=args.steps)
    os.makedirs(args.out, exist_ok=True)
    np.save(os.path.join(args.out, "samples.npy"), samples.cpu().numpy())
    print(f"wrote {samples.shape[0]} clouds of {samples.shape[1]} points to {args.out}")


if __name__ == "__main__":
    main()
