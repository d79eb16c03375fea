"""Dev tool: the latent sampler's step loop alone (LatentDiffusion._run pieces), T steps, eager v. graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import LatentDiffusion, Stepper
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
B, T = 32, int(os.environ.get("T", 1000))
for rep in range(2):
    for mode in ("eager", "graph1", "graph8", "graph32"):
        z = torch.randn(B, 256, device="cuda")
        tab = m.ddim_table(T, B)
        bias = m.model.time_bias(tab.t)
        stp = Stepper(m, z, tab, bias, m._forward_fn(), "ddim")
        stp.step(0, True)
        per = {"eager": 0, "graph1": 1, "graph8": 8, "graph32": 32}[mode]
        if per:
            stp.capture(per)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 1
        if per:
            while k + per <= T:
                stp.replay(); k += per
        while k < T:
            stp.step(k, True); k += 1
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep:
            print(f"{mode:8s}: {dt / (T - 1) * 1e6:6.1f} us/step (host issue done after {t_issue / (T - 1) * 1e6:5.1f} us/step)", flush=True)
