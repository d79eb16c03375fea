"""Shared test helpers: synthetic weights as torch tensors, tolerances."""
import numpy as np
import torch

import shapegen_amd  # noqa: F401
from shapegen_amd import specs

POINT_GAIN = 1.3
LATENT_GAIN = 1.3
VAE_GAIN = 1.3
ATTN_GAIN = 1.0


def as_torch(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def point_sd(prefix="model."):
    return as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix=prefix), seed=0, gain=POINT_GAIN))


def latent_sd():
    sd = specs.synth_state_dict(specs.latent_unet_spec(prefix="model."), seed=0, gain=LATENT_GAIN)
    sd.update(specs.synth_state_dict(specs.vae3d_large_spec(prefix="vae."), seed=0, gain=VAE_GAIN))
    return as_torch(sd)


def sab_sd(C):
    return as_torch(specs.synth_state_dict(specs.set_attention_spec(C), seed=C, gain=ATTN_GAIN))


def una_sd():
    return as_torch(specs.synth_state_dict(specs.unet_attention_spec(), seed=0, gain=ATTN_GAIN))


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def voxels_from_idx(idx_list):
    v = np.zeros((len(idx_list), 32 * 32 * 32), np.float32)
    for i, idx in enumerate(idx_list):
        v[i, idx] = 1
    return torch.from_numpy(v.reshape(len(idx_list), 1, 32, 32, 32))


def vae3d_small_sd():
    return as_torch(specs.synth_state_dict(specs.vae3d_small_spec(), seed=3, gain=VAE_GAIN))


def synth_voxels(b, seed):
    """(B,1,32,32,32) occupancy in {0,1}: three axis-aligned ellipsoids per grid from the integer hash of `specs`
    (the inputs `oracle/make_golden.py cfg4` fed the reference; `cfg4.npz` stores the occupancy counts as a check)."""
    v = np.zeros((b, 1, 32, 32, 32), np.float32)
    zz, yy, xx = np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing="ij")
    for i in range(b):
        u = specs.hash_uniform(f"vox{i}", 3 * 6, seed).reshape(3, 6) * 0.5 + 0.5
        for j in range(3):
            c = 6 + u[j, :3] * 20
            r = 3 + u[j, 3:] * 6
            m = ((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 <= 1
            v[i, 0][m] = 1
    return torch.from_numpy(v)
