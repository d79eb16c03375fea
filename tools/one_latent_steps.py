"""Dev tool: 30 eager latent denoiser steps at B=32 (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import LatentDiffusion, Stepper
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
z = torch.randn(32, 256, device="cuda")
tab = m.ddim_table(1000, 32)
stp = Stepper(m, z, tab, m.model.time_bias(tab.t), m._forward_fn(), "ddim")
for k in range(30):
    stp.step(k, True)
torch.cuda.synchronize()
