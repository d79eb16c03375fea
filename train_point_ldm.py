"""Entry point kept from the reference (reference train_point_ldm.py:150-234): VAE -> VAE samples -> latent
diffusion -> latent-diffusion samples, on the HIP VAE3DLarge / latent denoiser.  `train_vae` (train_point_ldm.py:24-79)
runs on `training_vae.VAETrainer` (BCE + annealed KL, Adam, plateau scheduler), `train_diffusion` (:81-110) on the HIP
latent trainer (frozen VAE encode -> L1 loss -> AdamW + cosine schedule).

    python train_point_ldm.py [--vae-ckpt vae.ckpt | --train-vae-epochs N] [--diffusion-ckpt ldm.ckpt | --train-diffusion-epochs N]
                              [--data-dir DIR] [--category table] [--steps 1000]
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

import shapegen_amd  # noqa: F401
from shapegen_amd import specs
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge as VAE


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vae-ckpt", default=None)
    ap.add_argument("--diffusion-ckpt", default=None)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--train-vae-epochs", type=int, default=0, help="train the voxel VAE first (train_point_ldm.py:24-79, `train_vae`)")
    ap.add_argument("--train-diffusion-epochs", type=int, default=0, help="0 = the reference default (perform_diffusion_training = False)")
    ap.add_argument("--data-dir", default=os.path.join("data", "shape_net_voxel_data_v1"))
    ap.add_argument("--category", default="table")
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--synthetic-shapes", type=int, default=160)
    ap.add_argument("--out", default=os.path.join("samples", "point_ldm"))
    args = ap.parse_args()
    torch.manual_seed(24)
    is_voxel_based = True                                     # train_point_ldm.py:160: VAE3DLarge path
    if args.vae_ckpt:
        vae = VAE.load_from_checkpoint(args.vae_ckpt)
    else:
        vae = VAE()
        sd = specs.synth_state_dict(specs.vae3d_large_spec(256), seed=2, gain=1.3)
        vae.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        print("no --vae-ckpt given: training is out of scope here, using synthetic VAE weights")
    vae = vae.to("cuda")
    if args.train_vae_epochs > 0:                            # train_point_ldm.py:24-79 (`train_vae`)
        from shapegen_amd.training import fit
        fit(vae, _data_module(args), max_epochs=args.train_vae_epochs, ckpt_dir=os.path.join("checkpoints", "point_ldm", "vae"), ckpt_name="vae")
    vae = vae.eval()
    os.makedirs(args.out, exist_ok=True)
    num_samples = 10
    samples_vae = vae.sample(num_samples=num_samples)                       # train_point_ldm.py:197
    np.savez_compressed(os.path.join(args.out, "vae_samples.npz"), **{f"sample_{i}": c.cpu().numpy() for i, c in enumerate(samples_vae)})
    print(f"Generated {num_samples} VAE samples")
    if args.diffusion_ckpt:
        diffusion = LatentDiffusion.load_from_checkpoint(args.diffusion_ckpt, vae=vae, is_voxel_based=is_voxel_based)
    else:
        diffusion = LatentDiffusion(vae, is_voxel_based=is_voxel_based)
        lsd = specs.synth_state_dict(specs.latent_unet_spec(prefix="model."), seed=1, gain=1.3)
        diffusion.load_state_dict({**{k: torch.from_numpy(np.asarray(v)) for k, v in lsd.items()},
                                   **{f"vae.{k}": v for k, v in vae.state_dict().items()}}, strict=True)
        print("no --diffusion-ckpt given: sampling the latent diffusion from synthetic weights")
    diffusion = diffusion.to("cuda")
    if args.train_diffusion_epochs > 0:                      # train_point_ldm.py:81-110 (`train_diffusion`)
        from shapegen_amd.training import fit
        fit(diffusion, _data_module(args), max_epochs=args.train_diffusion_epochs,
            ckpt_dir=os.path.join("checkpoints", "point_ldm", "latent_diffusion"), ckpt_name="latent_diffusion")
    diffusion = diffusion.eval()
    samples = diffusion.sample(num_samples=num_samples, num_steps=args.steps)  # train_point_ldm.py:222
    np.savez_compressed(os.path.join(args.out, "latent_diffusion_samples.npz"), **{f"sample_{i}": c.cpu().numpy() for i, c in enumerate(samples)})
    print(f"Generated {num_samples} diffusion denoised samples ({[int(c.shape[0]) for c in samples]} points)")


def _data_module(args):
    """Voxel batches: the reference's data module (train_point_ldm.py:166-168) or synthetic occupancy grids."""
    if os.path.isdir(args.data_dir):
        from shapegen_amd.data import PointCloudDataDirectoryModule
        return PointCloudDataDirectoryModule(args.data_dir, num_points=2048, batch_size=args.batch_size, file_mode="voxels",
                                             output_mode="voxels", augmentations=False, relevant_object_categories=[args.category])
    print(f"{args.data_dir} not found: training on {args.synthetic_shapes} synthetic occupancy grids")
    return _SyntheticVoxels(args.synthetic_shapes, args.batch_size)


class _SyntheticVoxels:
    """(B,1,32,32,32) occupancy batches of ellipsoid blobs, 80/20 train/val."""

    def __init__(self, count: int, batch_size: int, seed: int = 24):
        rng = np.random.default_rng(seed)
        zz, yy, xx = np.meshgrid(*[np.arange(32)] * 3, indexing="ij")
        v = np.zeros((count, 1, 32, 32, 32), np.float32)
        for i in range(count):
            c, r = rng.uniform(8, 24, (3, 3)), rng.uniform(3, 9, (3, 3))
            for j in range(3):
                v[i, 0][((zz - c[j, 0]) / r[j, 0]) ** 2 + ((yy - c[j, 1]) / r[j, 1]) ** 2 + ((xx - c[j, 2]) / r[j, 2]) ** 2 <= 1] = 1
        self.v, self.bs = torch.from_numpy(v), batch_size

    def setup(self):
        self.n_train = int(0.8 * len(self.v))

    def train_dataloader(self):
        return (self.v[i:i + self.bs] for i in range(0, self.n_train, self.bs))

    def val_dataloader(self):
        return (self.v[i:i + self.bs] for i in range(self.n_train, len(self.v), self.bs))


if __name__ == "__main__":
    main()
