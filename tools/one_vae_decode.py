"""Dev tool: a few VAE3DLarge decodes at B=32 (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import shapegen_amd
from shapegen_amd import specs
from shapegen_amd.vae import VAE3DLarge
torch.set_grad_enabled(False)
vae = VAE3DLarge()
sd = specs.synth_state_dict(specs.vae3d_large_spec(), seed=0, gain=1.3)
vae.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
vae = vae.to("cuda").eval()
z = torch.randn(32, 256, device="cuda")
x = (torch.rand(32, 1, 32, 32, 32, device="cuda") > 0.9).float()
for _ in range(3):
    y = vae.decode(z); mu, lv = vae.encode(x)
torch.cuda.synchronize()
