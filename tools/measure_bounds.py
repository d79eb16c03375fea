"""Dev tool: the measured values behind the parity bounds of the deeper networks (attention U-Net with its skip taps, VAE3DLarge) -- run on
the GPU box, then set each test bound at ~2x the measurement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, shapegen_amd
from helpers import una_sd, latent_sd, rel_l2, voxels_from_idx
from oracle import torch_oracle as O
from shapegen_amd.networks import UNetAttentionPointExperimental
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
torch.set_grad_enabled(False)
g = dict(np.load("tests/golden/attention.npz"))
sd = una_sd()
net = UNetAttentionPointExperimental(128); net.load_state_dict(sd, strict=True); net = net.to("cuda").eval()
x, t = torch.from_numpy(g["una_x"]), torch.from_numpy(g["una_t"])
eps = net(x.cuda(), t.cuda()).cpu()
taps = {}
want = O.unet_attention(sd, "", x, t, taps=taps)
print("attention U-Net (2,128): eps vs golden", rel_l2(eps, g["una_eps"]), "vs oracle", rel_l2(eps, want))
for k in ("x1", "x2", "x3"):
    print("   tap", k, rel_l2(net.tap(k, 2, 128).float().cpu(), taps[k]))
gen = torch.Generator().manual_seed(9)
x2, t2 = torch.randn(2, 2048, 3, generator=gen), torch.rand(2, generator=gen)
net2 = UNetAttentionPointExperimental(2048); net2.load_state_dict(sd, strict=True); net2 = net2.to("cuda").eval()
e2 = net2(x2.cuda(), t2.cuda()).cpu()
taps2 = {}
w2 = O.unet_attention(sd, "", x2, t2, taps=taps2)
print("attention U-Net (2,2048): eps vs oracle", rel_l2(e2, w2), [ (k, rel_l2(net2.tap(k, 2, 2048).float().cpu(), taps2[k])) for k in ("x1","x2","x3")])
gl = dict(np.load("tests/golden/latent.npz"))
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
vox = voxels_from_idx([gl["vae_occ_idx"], gl["vae_occ_idx1"]]).cuda()
mu, lv = m.vae.encode(vox)
print("VAE encode B=2: mu", rel_l2(mu.cpu(), gl["vae_mu"]), "logvar", rel_l2(lv.cpu(), gl["vae_logvar"]))
dec = m.vae.decode(torch.from_numpy(gl["vae_mu"]).cuda()).cpu()
err = (dec - torch.from_numpy(gl["vae_dec"])).abs()
print("VAE decode B=2: max", float(err.max()), "mean", float(err.mean()))
