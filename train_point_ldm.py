"""Entry point kept from the reference (reference train_point_ldm.py:150-234): VAE -> VAE samples -> latent
diffusion -> latent-diffusion samples, on the HIP VAE3DLarge / latent denoiser.  `train_vae` / `train_diffusion`
(Lightning `trainer.fit`) are outside this framework's scope: pass checkpoints.

    python train_point_ldm.py [--vae-ckpt vae.ckpt] [--diffusion-ckpt ldm.ckpt] [--steps 1000]
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
    vae = vae.to("cuda").eval()
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
    diffusion = diffusion.to("cuda").eval()
    samples = diffusion.sample(num_samples=num_samples, num_steps=args.steps)  # train_point_ldm.py:222
    np.savez_compressed(os.path.join(args.out, "latent_diffusion_samples.npz"), **{f"sample_{i}": c.cpu().numpy() for i, c in enumerate(samples)})
    print(f"Generated {num_samples} diffusion denoised samples ({[int(c.shape[0]) for c in samples]} points)")


if __name__ == "__main__":
    main()
