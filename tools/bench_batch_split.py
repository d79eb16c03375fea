"""Dev tool: does running the point U-Net on a quarter (half) of the batch at a time keep the activations in the 256-MB Infinity Cache?
One forward at B = 64, N = 2048 against 2 x B = 32 and 4 x B = 16 through the same workspace (MI355X only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
torch.set_grad_enabled(False)
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); t = torch.rand(64, device="cuda")

def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for rep in range(2):
    for parts in (1, 2, 4, 8):
        b = 64 // parts
        def run():
            for i in range(parts):
                model.model(x[i * b:(i + 1) * b], t[i * b:(i + 1) * b])
        ms = ev(run)
        if rep: print(f"{parts} x forward(B = {b}): {ms:.3f} ms", flush=True)
