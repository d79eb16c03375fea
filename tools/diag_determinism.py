"""Dev tool: bitwise repeatability screen of the point path at the headline shape (B = 64, N = 2048): forward of both backbones, the set-attention
block at every head width, the evaluation metrics.  No kernel on these paths has a run-dependent summation order (the column max is an integer
atomic max), so any run that differs from the first points at a staging race (tools/diag_vae_batch.py is the same screen for the VAE)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd import metrics
from helpers import point_sd
torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 2048, 3, generator=g).cuda(); tt = torch.rand(64, generator=g).cuda()
junk = torch.randn(4096, 4096, device="cuda")

def screen(name, fn, n=N):
    first = fn()
    first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else [first])]
    bad = 0
    for it in range(n):
        if it % 4 == 0: globals()["junk"] = junk @ junk * 1e-4
        out = fn()
        out = out if isinstance(out, (tuple, list)) else [out]
        if not all(torch.equal(a, b) for a, b in zip(out, first)): bad += 1
    print(f"{name}: {n} repetitions, {bad} differ bitwise from the first", flush=True)

model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
screen("UNetPointNetLarge forward, B=64 N=2048", lambda: model.model(x, tt))
screen("DDIM sample, 8 steps from a fixed x_T", lambda: model.sample(64, 2048, num_steps=8, x_T=x), n=max(N // 5, 10))
att = PointCloudDiffusion(num_points=2048, backbone="attention").to("cuda").eval()
screen("UNetAttentionPointExperimental forward, B=64 N=2048", lambda: att.model(x, tt))
a = torch.randn(16, 2048, 3, generator=g).cuda(); b = torch.randn(16, 2048, 3, generator=g).cuda()
screen("chamfer_distance, 16 pairs of 2048", lambda: metrics.chamfer_distance(a, b))
