import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, shapegen_amd
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
g = dict(np.load("tests/golden/cfg4.npz"))
rows = g["dec_rows"]
for name, zz, want in (("z0", g["z0"], g["dec"]), ("mu", g["enc_mu"], g["dec_of_mu"])):
    dec = m.vae.decode(torch.from_numpy(zz).cuda())[torch.from_numpy(rows).cuda()].cpu()
    want = torch.from_numpy(want).float()
    e = (dec - want).abs().flatten()
    q = torch.quantile(e[::7].double(), torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], dtype=torch.float64))
    flips = ((dec > 0.4) != (want > 0.4)).float().mean()
    print(name, "|z| max", float(np.abs(zz).max()), "err max", float(e.max()), "mean", float(e.mean()), "quantiles", [f"{v:.2e}" for v in q.tolist()],
          "flips@0.4", float(flips), "frac>2e-2", float((e > 2e-2).float().mean()))
    # where the big errors are: probability of the reference there
    big = e > 2e-2
    if big.any():
        w = want.flatten()[big]
        print("   reference p at big-error voxels: min", float(w.min()), "max", float(w.max()), "mean", float(w.mean()))
