"""Dev tool: latent DDIM loop, B = 32 (env B: up to 64 = two interleaved streams), T = 1000: one persistent launch (csrc/latent_persist.hip) against the per-layer
launches replayed as 8-step graphs, A/B in one process; optional poll back-off sweep (SLEEPS="0 1 2 4")."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
lib = _lib.load()
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 1000))
zT = torch.randn(B, 256, device="cuda")

def loop(persistent):
    m.use_persistent = persistent
    z = m._start(B, zT)
    tab = m.ddim_table(T, B)
    bias = m.model.time_bias(tab.t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z0 = m._run(z, tab, bias, m._forward_fn(), "ddim")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / T * 1e6, z0

for rep in range(3):
    for persistent in (True, False):
        us, z0 = loop(persistent)
        if rep:
            print(f"persistent={int(persistent)}: {us:6.1f} us/step  |z0| {float(z0.abs().max()):.3f}", flush=True)
for sl in [int(v) for v in os.environ.get("SLEEPS", "").split()]:
    h, _ = m.model._persist_handle()
    for pred in (0, 1, 2, 3):
        lib.pcd_latent_persist_config(h, sl, pred)
        us = min(loop(True)[0] for _ in range(3))
        print(f"poll sleep {sl} predict {pred}: {us:6.1f} us/step", flush=True)
lib.pcd_latent_persist_config(m.model._persist_handle()[0], 0, 1)
