"""Dev tool: point U-net (cfg2: B=64, N=2048) with the chained narrow layers on and off, in one process:
max |eps difference| of one forward, then eager steps/s of the DDPM step, alternating A/B."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import PointCloudDiffusion, Stepper
from helpers import point_sd
torch.set_grad_enabled(False)
lib = _lib.load()
B, N = 64, 2048
m = PointCloudDiffusion(num_points=N); m.load_state_dict(point_sd(), strict=True); m = m.to("cuda").eval()
x = torch.randn(B, N, 3, device="cuda"); t = torch.randint(0, 1000, (B,), device="cuda")
outs = []
for ch in (0, 1):
    lib.pcd_unet_config(ch)
    outs.append(m.model(x, t).float().clone())
d = (outs[0] - outs[1]).abs().max().item(); ref = outs[0].abs().max().item()
print(f"forward: max|chain - layers| = {d:.3e} (max |eps| {ref:.3f}), rel-L2 {((outs[0]-outs[1]).norm()/outs[0].norm()).item():.3e}", flush=True)
tab = m.ddpm_table(1000, B); bias = m.model.time_bias(tab.t)
for rep in range(3):
    for ch in (1, 0):
        lib.pcd_unet_config(ch)
        xx = m._randn_like(torch.empty(B, N, 3, device="cuda"))
        stp = Stepper(m, xx, tab, bias, m._forward_fn(), "ddpm")
        for k in range(5):
            stp.step(k, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(5, 65):
            stp.step(k, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"chains={ch}: {60 / dt:6.1f} steps/s  ({dt / 60 * 1e3:.3f} ms/step)", flush=True)
lib.pcd_unet_config(1)
