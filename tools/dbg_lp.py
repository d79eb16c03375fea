import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
lib = _lib.load()
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
zT = torch.randn(32, 256, device="cuda")
h, _ = m.model._persist_handle()
for pred in (0, 1):
    for T in (2, 3, 8, 100):
        lib.pcd_latent_persist_config(h, 1, pred)
        try:
            _, z0 = m.sample(32, num_steps=T, z_T=zT, return_latent=True)
            print(f"predict {pred} T {T}: ok |z0| {float(z0.abs().max()):.3f}", flush=True)
        except RuntimeError as e:
            print(f"predict {pred} T {T}: {str(e)[:90]}", flush=True)
