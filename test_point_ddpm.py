"""Evaluation entry point with the reference's surface (reference test_point_ddpm.py):
`test_ddpm_generation` (:24-56) and `test_ddpm_reconstruction` (:58-119), running the HIP sampler.

    python test_point_ddpm.py [--ckpt-dir DIR] [--num-samples 16] [--num-points 2048] [--steps 1000]

Without checkpoints (none ship with the reference) it evaluates a model with deterministic
synthetic weights on synthetic ShapeNet-shaped clouds, so the plumbing (BASELINE configs[0]) is
exercised end to end.  Multi-GPU: launch with torch.distributed.run; samples are sharded across
ranks and metric rows are all-gathered (RCCL).
"""
from __future__ import annotations

import argparse
import glob
import logging
import os

import numpy as np
import torch

import shapegen_amd  # noqa: F401
from shapegen_amd import dist as D
from shapegen_amd import specs
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd.utils import setup_logger

LOG = "test_logger_point_ddpm"


def synthetic_clouds(batch: int, num_points: int, seed: int = 24) -> torch.Tensor:
    """Grid-like clouds shaped like data.py:213-254 output: voxel coords, centred, unit radius."""
    rng = np.random.default_rng(seed)
    out = np.zeros((batch, num_points, 3), np.float32)
    for i in range(batch):
        c = rng.uniform(8, 24, (3, 3))
        r = rng.uniform(3, 9, (3, 3))
        zz, yy, xx = np.meshgrid(*[np.arange(32)] * 3, indexing="ij")
        occ = np.zeros((32, 32, 32), bool)
        for j in range(3):
            occ |= ((zz - c[j, 0]) / r[j, 0]) ** 2 + ((yy - c[j, 1]) / r[j, 1]) ** 2 + ((xx - c[j, 2]) / r[j, 2]) ** 2 <= 1
        pts = np.stack(np.where(occ), 1).astype(np.float32)
        pts -= pts.mean(0)
        pts /= np.linalg.norm(pts, axis=1).max()
        idx = rng.choice(len(pts), num_points, replace=len(pts) < num_points)
        out[i] = pts[idx]
    return torch.from_numpy(out)


def test_ddpm_generation(model, model_name, num_samples=10, num_points=2048, num_steps=1000):
    """reference test_point_ddpm.py:24-56 (DDIM `sample`); returns the generated clouds."""
    with torch.no_grad():
        generated = D.sample_sharded(model, num_samples, num_points, num_steps)   # rank-disjoint start noise
    logging.getLogger(LOG).info(f"Generated {generated.shape[0]} samples for {model_name}.")
    return generated


def test_ddpm_reconstruction(model, model_name, original_samples, initial_t=0.010, num_steps=1000):
    """reference test_point_ddpm.py:58-119: add_noise(t) -> sample3 -> per-sample metrics."""
    rank, world = D.world()
    lo, hi = D.shard_range(original_samples.shape[0], rank, world)
    orig = original_samples[lo:hi].to(model.device)
    with torch.no_grad():
        t = torch.ones(orig.shape[0], device=model.device) * initial_t
        with D.shard_context(model, lo, original_samples.shape[0]):
            noisy, _, _, _ = model.add_noise(orig, t)
        recon = model.sample3(num_samples=orig.shape[0], num_points=orig.shape[1], x=noisy, start_t=t, num_steps=num_steps)
        rows, mean = D.evaluate_sharded(orig, recon)
    log = logging.getLogger(LOG)
    log.info(f"Average Chamfer Distance: {float(mean[0]):.3f}")
    log.info(f"Average Earth Mover's Distance: {float(mean[1]):.3f}")
    log.info(f"Average Reconstruction Loss: {float(mean[2]):.3f}")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt-dir", default=os.path.join("checkpoints", "best_run", "point_cloud_diffusion"))
    ap.add_argument("--num-samples", type=int, default=16)
    ap.add_argument("--num-points", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join("test", "outputs"))
    args = ap.parse_args()
    torch.manual_seed(24)                                  # pl.seed_everything(24), test_point_ddpm.py:13
    rank, world, local = D.init_from_env()
    device = torch.device("cuda", local)
    setup_logger(LOG, os.path.join("test", "logs", "point_ddpm_test.log"))
    ckpts = sorted(glob.glob(os.path.join(args.ckpt_dir, "*.ckpt")))
    models = []
    for path in ckpts:
        models.append((os.path.basename(path)[:-5], PointCloudDiffusion.load_from_checkpoint(path)))
    if not models:
        m = PointCloudDiffusion(num_points=args.num_points)
        sd = specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=1.3)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        models.append(("synthetic_weights", m))
    val = synthetic_clouds(args.num_samples, args.num_points)
    os.makedirs(args.out, exist_ok=True)
    for name, model in models:
        model = model.to(device).eval()
        gen = test_ddpm_generation(model, name, args.num_samples, args.num_points, args.steps)
        rows = test_ddpm_reconstruction(model, name, val, num_steps=args.steps)
        if rank == 0:
            np.savez_compressed(os.path.join(args.out, f"{name}.npz"), generated=gen.cpu().numpy(), metrics=rows.cpu().numpy())


if __name__ == "__main__":
    main()
