"""Dev tool: layer-by-layer comparison of the HIP training forward with the CPU oracle in train() mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import shapegen_amd
from helpers import point_sd, rel_l2
from oracle import torch_oracle as O
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd.training import PointTrainer

g = np.load("tests/golden/train.npz")
sd = point_sd()
model = PointCloudDiffusion(num_points=128); model.load_state_dict(sd, strict=True); model = model.to("cuda")
x_t, t, noise = (torch.from_numpy(g[k]) for k in ("x_t", "t", "noise"))
tr = PointTrainer(model.model)
pred = tr.forward(x_t.cuda(), t.cuda(), update_stats=False)
taps = {}
sd_ref = {k: v.clone() for k, v in sd.items()}
ref = O.unet_pointnet_large(sd_ref, "model.", x_t, t, taps=taps, train=True)
B, N = 2, 128
def cmp(name, mine, want):
    want = want.transpose(2, 1).reshape(B * N, -1)
    print(f"{name:8s} rel_l2 {rel_l2(mine.float().cpu(), want):.3e}  |ref| {want.abs().mean():.3f}")
print("temb", rel_l2(tr.temb.cpu(), taps["temb"]))
cmp("x1", tr.enc[0][2].a, taps["x1"]); cmp("x2", tr.enc[1][2].a, taps["x2"]); cmp("x3", tr.enc[2][2].a, taps["x3"]); cmp("x4", tr.enc[3][2].a, taps["x4"])
print("pooled", rel_l2(tr.gmax.cpu(), taps["pooled"]))
cmp("d4", tr.dec[0][2].a, taps["d4"]); cmp("d3", tr.dec[1][2].a, taps["d3"]); cmp("d2", tr.dec[2][2].a, taps["d2"]); cmp("d1", tr.dec[3][2].a, taps["d1"])
print("pred", rel_l2(pred.cpu(), ref))
