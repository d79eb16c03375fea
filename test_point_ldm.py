"""VAE evaluation entry point with the reference's surface (reference test_point_ldm.py): `test_vae_generation`
(:23-56) and `test_vae_reconstruction` (:58-119) on the HIP VAE3DLarge.

    python test_point_ldm.py [--ckpt-dir DIR] [--data-dir DIR] [--category table] [--num-samples 16] [--threshold 0.5]

Checkpoints are read with the Lightning-free loader; the validation grids come from `shapegen_amd.data` (the
reference's PointCloudDataDirectoryModule contract: file_mode='voxels', output_mode='voxels', batch 16).  With no
checkpoint / data directory present (none ship with the reference) it runs on deterministic synthetic weights and
synthetic occupancy grids so the plumbing is exercised end to end.  Multi-GPU: launch with torch.distributed.run;
grids are sharded across ranks and the per-sample metric rows are all-gathered.
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
from shapegen_amd.metrics import compute_metrics
from shapegen_amd.utils import setup_logger, voxel_tensor_to_point_clouds
from shapegen_amd.vae import VAE3DLarge as VAE3D

LOG = "test_logger_point_ldm"


def synthetic_voxels(batch: int, seed: int = 24) -> torch.Tensor:
    """(B,1,32,32,32) occupancy in {0,1}: three ellipsoid blobs per grid, ~5-15 % filled."""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(32)] * 3, indexing="ij")
    out = np.zeros((batch, 1, 32, 32, 32), np.float32)
    for i in range(batch):
        c, r = rng.uniform(8, 24, (3, 3)), rng.uniform(3, 9, (3, 3))
        for j in range(3):
            out[i, 0][((zz - c[j, 0]) / r[j, 0]) ** 2 + ((yy - c[j, 1]) / r[j, 1]) ** 2 + ((xx - c[j, 2]) / r[j, 2]) ** 2 <= 1] = 1
    return torch.from_numpy(out)


def test_vae_generation(model, model_name, num_samples=10, threshold=0.5):
    """reference test_point_ldm.py:23-56: decode prior samples -> list of (n_i, 3) clouds."""
    with torch.no_grad():
        lo, hi = D.shard_range(num_samples, *D.world())
        with D.shard_context(model, lo, num_samples):                # rank-disjoint prior draws
            generated = D.all_gather_clouds(model.sample(num_samples=hi - lo, threshold=threshold))
    logging.getLogger(LOG).info(f"Generated and saved {len(generated)} samples.")
    return generated


def test_vae_reconstruction(model, model_name, original_samples, num_samples=10, threshold=0.5):
    """reference test_point_ldm.py:58-119: grids -> model(x) -> both sides to point clouds -> per-sample metrics."""
    rank, world = D.world()
    original_samples = original_samples[:num_samples]
    lo, hi = D.shard_range(original_samples.shape[0], rank, world)
    with torch.no_grad():
        vox = original_samples[lo:hi].to(model.device)
        with D.shard_context(model, lo, original_samples.shape[0]):  # rank-disjoint reparameterisation draws
            recon_vox, _, _ = model(vox)
        orig = voxel_tensor_to_point_clouds(vox, threshold)
        recon = voxel_tensor_to_point_clouds(recon_vox, threshold)
        rows, _ = D.evaluate_sharded(orig, recon)                    # an empty cloud has no metrics: NaN row
    log = logging.getLogger(LOG)
    mean = torch.nanmean(rows, dim=0)
    log.info(f"Average Chamfer Distance: {float(mean[0]):.3f}")
    log.info(f"Average Earth Mover's Distance: {float(mean[1]):.3f}")
    log.info(f"Average Reconstruction Loss: {float(mean[2]):.3f}")
    log.info(f"Reconstructed and saved {rows.shape[0]} samples.")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--category", default="table")
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--data-dir", default=os.path.join("data", "shape_net_voxel_data_v1"))
    ap.add_argument("--num-samples", type=int, default=16)
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--out", default=os.path.join("test", "outputs"))
    args = ap.parse_args()
    torch.manual_seed(24)                                    # pl.seed_everything(24), test_point_ldm.py:12
    rank, world, local = D.init_from_env()
    device = torch.device("cuda", local)
    setup_logger(LOG, os.path.join("test", "logs", "point_ldm_test.log"))
    sub = f"{args.category}_from_scratch_no_augs_voxel_simoid_bce_kl_mean_beta_warmup_annealed_upto_100"
    ckpt_dir = args.ckpt_dir or os.path.join("checkpoints", "best_run", "point_ldm", sub)
    models = [(f"{sub}-{os.path.basename(p)[:-5]}", VAE3D.load_from_checkpoint(p)) for p in sorted(glob.glob(os.path.join(ckpt_dir, "*.ckpt")))]
    if not models:
        m = VAE3D()
        sd = specs.synth_state_dict(specs.vae3d_large_spec(256), seed=2, gain=1.3)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        models.append(("synthetic_weights", m))
    if os.path.isdir(args.data_dir):
        from shapegen_amd.data import PointCloudDataDirectoryModule
        dm = PointCloudDataDirectoryModule(args.data_dir, num_points=2048, batch_size=16, file_mode="voxels", output_mode="voxels",
                                           augmentations=False, relevant_object_categories=[args.category])
        dm.setup()
        val = next(iter(dm.val_dataloader()))
    else:
        val = synthetic_voxels(args.num_samples)
    os.makedirs(args.out, exist_ok=True)
    for name, model in models:
        model = model.to(device).eval()
        gen = test_vae_generation(model, name, num_samples=args.num_samples, threshold=args.threshold)
        rows = test_vae_reconstruction(model, name, val, num_samples=args.num_samples, threshold=args.threshold)
        if rank == 0:
            np.savez_compressed(os.path.join(args.out, f"vae_{name}.npz"), metrics=rows.cpu().numpy(),
                                **{f"sample_{i}": c.cpu().numpy() for i, c in enumerate(gen)})


if __name__ == "__main__":
    main()
