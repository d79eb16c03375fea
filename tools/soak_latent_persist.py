"""Dev tool: soak of the persistent latent kernel: many 1000-step DDIM calls at B = 32 (one stream) and B = 64 / 45 (two interleaved
streams); every call must report status 0 and reproduce the first call bitwise (a stale or torn exchange read would show as a difference)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
m.use_persistent = True
reps = int(os.environ.get("REPS", 40))
for b in (32, 64, 45, 7):
    zT = torch.randn(b, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(b))
    first = None
    for i in range(reps):
        _, z0 = m.sample(b, num_steps=1000, z_T=zT, return_latent=True)
        m.model.check_persist_status()
        if first is None: first = z0.clone()
        assert torch.equal(z0, first), (b, i)
    print(f"B = {b}: {reps} x 1000 steps bitwise reproducible, finite: {bool(torch.isfinite(first).all())}", flush=True)
