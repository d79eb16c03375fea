"""Dev tool: where does the fp16 product path drift from the reference over the 1000-step horizon?
Runs the three samplers at (2 | 4, 2048) from the G19-G21 start states in BOTH arithmetic modes, eagerly, and prints
(a) the state's rel-L2 between the fp16 path and the fp32 parity mode every 50 steps, and (b) both against the reference's
recorded states at the golden checkpoints (the denoiser's input at calls 0, 100, 250, 500, 750, 900, 990, 999).
    python tools/divergence_t1000.py > profiles/r04_b_t1000_divergence.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from helpers import point_sd, rel_l2  # noqa: E402
from shapegen_amd import specs  # noqa: E402
from shapegen_amd.diffusion import PointCloudDiffusion  # noqa: E402

torch.set_grad_enabled(False)
G = os.path.join(ROOT, "tests", "golden")


GAIN = [1.3]


def build(prec):
    m = PointCloudDiffusion(num_points=2048)
    from helpers import as_torch
    m.load_state_dict(as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=GAIN[0])), strict=True)
    m = m.to("cuda").eval()
    m.model.set_precision(prec)
    m.use_graphs = False
    return m


def record(m, run):
    states = []
    inner = m.model.forward_with_bias

    def fwd(x, tb, stride, out=None):
        states.append(x.detach().clone())
        return inner(x, tb, stride, out=out)

    m.model.forward_with_bias = fwd
    try:
        out = run(m)
    finally:
        m.model.forward_with_bias = inner
    return states, out


class Hashed:
    def __init__(self, tag, shape):
        self.tag, self.shape = tag, shape

    def __getitem__(self, k):
        return torch.from_numpy(specs.hash_normal(f"{self.tag}{k}", int(np.prod(self.shape)), 0).astype(np.float32).reshape(self.shape))


def report(name, g, run):
    s16, o16 = record(build("fp16"), run)
    s32, o32 = record(build("fp32"), run)
    print(f"== {name}: final rel-L2 fp16 vs reference {rel_l2(o16.cpu(), g['out']):.3e}, fp32 vs reference {rel_l2(o32.cpu(), g['out']):.3e}, "
          f"fp16 vs fp32 {rel_l2(o16.cpu(), o32.cpu()):.3e}; max-abs fp32 vs reference {float((o32.cpu() - torch.from_numpy(g['out'])).abs().max()):.3e}")
    print("   call   |state|rms   fp16 vs fp32 (rel-L2)")
    for k in list(range(0, 1000, 50)) + [990, 999]:
        print(f"   {k:4d}   {float(s32[k].pow(2).mean().sqrt()):9.4f}   {rel_l2(s16[k].cpu(), s32[k].cpu()):.3e}")
    print("   call   fp16 vs reference   fp32 vs reference   (rel-L2 of the state handed to the denoiser)")
    rows = g["ckpt_x"].shape[1]
    for i, c in enumerate(g["ckpt_calls"]):
        ref = torch.from_numpy(g["ckpt_x"][i])
        print(f"   {int(c):4d}   {rel_l2(s16[int(c)][:rows].cpu(), ref):.3e}           {rel_l2(s32[int(c)][:rows].cpu(), ref):.3e}")


g = dict(np.load(os.path.join(G, "point_t1000_ddim.npz")))
report("DDIM sample(2, 2048), 1000 steps (G19)", g, lambda m: m.sample(2, 2048, x_T=torch.from_numpy(g["xT"]).cuda()))
GAIN[0] = 1.0
g2 = dict(np.load(os.path.join(G, "point_t1000_ddpm_stable.npz")))
report("DDPM sample2(2, 2048), 1000 steps, hashed per-step noise, weights at gain 1.0 (G20b)", g2,
       lambda m: m.sample2(2, 2048, x_T=torch.from_numpy(g2["xT"]).cuda(), noises=Hashed("g20.z", (2, 2048, 3))))
GAIN[0] = 1.3
g2r = dict(np.load(os.path.join(G, "point_t1000_ddpm.npz")))
report("DDPM sample2(2, 2048), 1000 steps, weights at gain 1.3: the reference's loop runs away to |x| = 9e8 (G20, runaway record)", g2r,
       lambda m: m.sample2(2, 2048, x_T=torch.from_numpy(g2r["xT"]).cuda(), noises=Hashed("g20.z", (2, 2048, 3))))
if os.path.exists(os.path.join(G, "point_t1000_recon.npz")):
    g3 = dict(np.load(os.path.join(G, "point_t1000_recon.npz")))
    t = torch.ones(4, device="cuda") * 0.010
    report("reconstruction sample3(4, 2048) from t = 0.01, 1000 steps (G21)", g3,
           lambda m: m.sample3(num_samples=4, num_points=2048, x=torch.from_numpy(g3["noisy"]).cuda(), start_t=t))
