"""Dev tool: determinism screen of VAE3DLarge.encode / decode (a staging race in a convolution kernel would show as a run that differs bitwise from the first):
N repetitions at B = 16 and B = 1 with the round-4 kernels on, then off, other work interleaved to vary the timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd, rel_l2
torch.set_grad_enabled(False)
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
g = torch.Generator().manual_seed(21)
z = torch.randn(16, 256, generator=g).cuda()
junk = torch.randn(4096, 4096, device="cuda")
for cfg in (14, 8, 4, 0):
    lib.pcd_vae_config(cfg)
    dec0 = m.vae.decode(z).clone()
    vox = (dec0 > 0.5).float()
    mu0, _ = m.vae.encode(vox); mu0 = mu0.clone()
    mu1_0, _ = m.vae.encode(vox[:1]); mu1_0 = mu1_0.clone()
    bad = {"decode16": 0, "encode16": 0, "encode1": 0}
    worst = 0.0
    for it in range(N):
        if it % 3 == 0: junk = junk @ junk * 1e-4                     # unrelated load in between
        d = m.vae.decode(z)
        if not torch.equal(d, dec0): bad["decode16"] += 1
        mu, _ = m.vae.encode(vox)
        if not torch.equal(mu, mu0): bad["encode16"] += 1; worst = max(worst, rel_l2(mu.cpu(), mu0.cpu()))
        mu1, _ = m.vae.encode(vox[:1])
        if not torch.equal(mu1, mu1_0): bad["encode1"] += 1; worst = max(worst, rel_l2(mu1.cpu(), mu1_0.cpu()))
    print(f"pcd_vae_config({cfg}): {N} repetitions, runs that differ bitwise from the first: {bad}, worst rel-L2 {worst:.2e}", flush=True)
lib.pcd_vae_config(1)
