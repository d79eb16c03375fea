"""Dev tool: which fp16 rounding carries the 1000-step DDPM drift of the product path -- the weights (a fixed perturbation of the
network, the same at every step) or the activations (a fresh rounding per layer and step)?  Runs G20b's loop (gain 1.0, hashed noise) in
three forms: fp32 mode (= the reference to ~1e-6), fp32 mode with the FOLDED weights rounded to fp16 first, and the fp16 product path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from helpers import rel_l2, as_torch  # noqa: E402
from shapegen_amd import packing, specs  # noqa: E402
from shapegen_amd.diffusion import PointCloudDiffusion  # noqa: E402

torch.set_grad_enabled(False)
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "point_t1000_ddpm_stable.npz")))
sd = as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=1.0))


class Hashed:
    def __getitem__(self, k):
        return torch.from_numpy(specs.hash_normal(f"g20.z{k}", 2 * 2048 * 3, 0).astype(np.float32).reshape(2, 2048, 3))


def diffuse(w):
    """fp16 rounding with the rounding error carried along K (error diffusion): every row's SUM of weights is kept to one ulp."""
    w = np.asarray(w, np.float64)
    q = np.empty_like(w)
    carry = np.zeros(w.shape[0])
    for k in range(w.shape[1]):
        v = w[:, k] + carry
        q[:, k] = v.astype(np.float16).astype(np.float64)
        carry = v - q[:, k]
    return q


def run(prec, round_weights=False):
    orig = packing.pack_point_unet
    if round_weights:
        f = diffuse if round_weights == "diffuse" else (lambda w: w.astype(np.float16).astype(np.float64))

        def rounded(*a, **k):
            lin, ex = orig(*a, **k)
            lin = [(f(w), b) for w, b in lin]
            ex["wg"] = f(ex["wg"])
            return lin, ex
        packing.pack_point_unet = rounded
    try:
        m = PointCloudDiffusion(num_points=2048)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda").eval()
        m.model.set_precision(prec)
        out = m.sample2(2, 2048, x_T=torch.from_numpy(g["xT"]).cuda(), noises=Hashed())
    finally:
        packing.pack_point_unet = orig
    return out.cpu()


ref = torch.from_numpy(g["out"])
a = run("fp32")
b = run("fp32", round_weights=True)
c = run("fp16")
d = run("fp32", round_weights="diffuse")
e = run("fp16", round_weights="diffuse")
print(f"fp32 mode vs reference:                         rel-L2 {rel_l2(a, ref):.3e}")
print(f"fp32 activations, fp16-rounded weights vs ref:  rel-L2 {rel_l2(b, ref):.3e}")
print(f"fp16 product path vs reference:                 rel-L2 {rel_l2(c, ref):.3e}")
print(f"fp16 product path vs fp16-weights/fp32-acts:    rel-L2 {rel_l2(c, b):.3e}")
print(f"fp32 activations, error-diffusion fp16 weights:   rel-L2 {rel_l2(d, ref):.3e}")
print(f"fp16 product path, error-diffusion fp16 weights:  rel-L2 {rel_l2(e, ref):.3e}")
