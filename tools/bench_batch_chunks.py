"""Dev tool: the point U-Net's forward at cfg2 (B = 64, N = 2048) as one launch sequence against the same batch in chunks of 32 / 16 / 8 shapes
(per-chunk activations of 268 / 134 / 67 MB at the widest layer: does a chunk that fits the 256-MB memory-side cache pay?)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
torch.set_grad_enabled(False)
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 2048, 3, generator=g).cuda(); tt = torch.rand(64, generator=g).cuda()
def ev(fn, n=20):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(3):
    for c in (64, 32, 16, 8):
        xs, ts = x.split(c), tt.split(c)
        t = ev(lambda: [model.model(a, b) for a, b in zip(xs, ts)])
        print(f"chunks of {c:2d} shapes: {t:.3f} ms per 64-shape forward", flush=True)
