"""Dev tool: latent step time (graph of 8 steps, B=32) with the LDS-DMA operand staging of the split-K GEMM on and off, A/B in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import LatentDiffusion, Stepper
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
lib = _lib.load()
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
B, T = 32, 1001
for rep in range(3):
    for dma in (1, 0):
        lib.pcd_skinny_config(dma)
        z = torch.randn(B, 256, device="cuda")
        tab = m.ddim_table(T, B)
        stp = Stepper(m, z, tab, m.model.time_bias(tab.t), m._forward_fn(), "ddim")
        stp.step(0, True)
        stp.capture(8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(125):
            stp.replay()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep:
            print(f"dma={dma}: {dt / 1000 * 1e6:6.1f} us/step", flush=True)
lib.pcd_skinny_config(1)
