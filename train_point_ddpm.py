"""Training entry point with the reference's surface (reference train_point_ddpm.py:25-99): build the data module
(voxel files -> 2048-point clouds, batch 16), construct or load `PointCloudDiffusion`, train it, then draw 10 samples.
Training runs on the HIP trainer (`shapegen_amd.training`): fp16 MFMA GEMMs for every forward / backward product,
BatchNorm batch statistics, L1 loss, AdamW + ReduceLROnPlateau, top-k checkpoints by val_loss in the reference's
`.ckpt` layout.

    python train_point_ddpm.py [--data-dir DIR] [--category chair] [--epochs 500] [--ckpt resume.ckpt] [--max-steps N]

Without a data directory (none ships with the reference) it trains on synthetic ShapeNet-shaped clouds so the whole
loop can be exercised.
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import numpy as np
import torch

import shapegen_amd  # noqa: F401
from shapegen_amd.data import PointCloudDataDirectoryModule, PointCloudDataModule
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd.training import fit
from shapegen_amd.utils import setup_logger


def synthetic_clouds(count: int, num_points: int, seed: int = 24) -> np.ndarray:
    """Grid-like clouds shaped like data.py:213-254 output: voxel coordinates of blobs, centred, unit radius."""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(32)] * 3, indexing="ij")
    out = np.zeros((count, num_points, 3), np.float32)
    for i in range(count):
        c, r = rng.uniform(8, 24, (3, 3)), rng.uniform(3, 9, (3, 3))
        occ = np.zeros((32, 32, 32), bool)
        for j in range(3):
            occ |= ((zz - c[j, 0]) / r[j, 0]) ** 2 + ((yy - c[j, 1]) / r[j, 1]) ** 2 + ((xx - c[j, 2]) / r[j, 2]) ** 2 <= 1
        pts = np.stack(np.where(occ), 1).astype(np.float32)
        pts -= pts.mean(0)
        pts /= np.linalg.norm(pts, axis=1).max()
        out[i] = pts[rng.choice(len(pts), num_points, replace=len(pts) < num_points)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None, help="reference-layout .ckpt to resume from")
    ap.add_argument("--data-dir", default=os.path.join("data", "shape_net_voxel_data_v1"))
    ap.add_argument("--category", default="chair")
    ap.add_argument("--num-points", type=int, default=2048)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--synthetic-shapes", type=int, default=160)
    ap.add_argument("--sample-steps", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join("samples", "point_cloud_diffusion"))
    args = ap.parse_args()
    torch.manual_seed(24)
    timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    logger = setup_logger("train_point_ddpm", os.path.join("train", "logs", f"train_point_ddpm_log_{timestamp}.log"))
    if os.path.isdir(args.data_dir):
        dm = PointCloudDataDirectoryModule(args.data_dir, num_points=args.num_points, batch_size=args.batch_size, file_mode="voxels",
                                           output_mode="point_clouds", augmentations=False,
                                           relevant_object_categories=[args.category])
    else:
        logger.info(f"{args.data_dir} not found: training on {args.synthetic_shapes} synthetic clouds")
        dm = _Unwrap(PointCloudDataModule(synthetic_clouds(args.synthetic_shapes, args.num_points), batch_size=args.batch_size))
    if args.ckpt:
        logger.info(f"Loading Diffusion model from checkpoint: {args.ckpt}")
        model = PointCloudDiffusion.load_from_checkpoint(args.ckpt)
        assert model.num_points == args.num_points
    else:
        model = PointCloudDiffusion(num_points=args.num_points)
    model = model.to("cuda")
    logger.info("Starting Diffusion Training")
    fit(model, dm, max_epochs=args.epochs, ckpt_dir=os.path.join("checkpoints", "point_ddpm", timestamp), log=logger.info,
        max_steps=args.max_steps)
    model.eval()
    samples = model.sample(num_samples=10, num_points=args.num_points, num_steps=args.sample_steps)   # train_point_ddpm.py:91-93
    os.makedirs(args.out, exist_ok=True)
    np.save(os.path.join(args.out, "samples.npy"), samples.cpu().numpy())
    print(f"wrote {samples.shape[0]} clouds of {samples.shape[1]} points to {args.out}")


class _Unwrap:
    """PointCloudDataModule yields 1-tuples (TensorDataset, data.py:32); the training loop wants the tensor."""

    def __init__(self, dm):
        self.dm = dm

    def setup(self):
        self.dm.setup()

    def train_dataloader(self):
        return (b[0] for b in self.dm.train_dataloader())

    def val_dataloader(self):
        return (b[0] for b in self.dm.val_dataloader())


if __name__ == "__main__":
    main()
