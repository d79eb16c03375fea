"""Dev tool: which layers' fp16 weight rounding carries the 1000-step DDPM drift (G20b)?  fp32 activations throughout (the fp32 parity mode);
only the named subset of the folded weight matrices is rounded to fp16 first."""
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
ref = torch.from_numpy(g["out"])


class Hashed:
    def __getitem__(self, k):
        return torch.from_numpy(specs.hash_normal(f"g20.z{k}", 2 * 2048 * 3, 0).astype(np.float32).reshape(2, 2048, 3))


def run(subset, wg, act_mask=0):
    orig = packing.pack_point_unet

    def rounded(*a, **k):
        lin, ex = orig(*a, **k)
        lin = [((w.astype(np.float16).astype(np.float64) if i in subset else w), b) for i, (w, b) in enumerate(lin)]
        if wg:
            ex["wg"] = ex["wg"].astype(np.float16).astype(np.float64)
        return lin, ex
    packing.pack_point_unet = rounded
    try:
        m = PointCloudDiffusion(num_points=2048)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda").eval()
        m.model.set_precision("fp32")
        if act_mask:
            from shapegen_amd import _lib
            m.model._ensure_packed()
            _lib.check(_lib.load().pcd_unet_f32_round_activations(m.model._handle, act_mask))
        out = m.sample2(2, 2048, x_T=torch.from_numpy(g["xT"]).cuda(), noises=Hashed())
    finally:
        packing.pack_point_unet = orig
    return rel_l2(out.cpu(), ref)


cases = (("none", set(), False), ("all", set(range(26)), True),
         ("enc1.conv2 (0)", {0}, False), ("enc1.conv3 (1)", {1}, False), ("enc2.conv1-2 (2,3)", {2, 3}, False), ("enc2.conv3 (4)", {4}, False),
         ("dec2 (19-21)", {19, 20, 21}, False), ("dec1.conv1 (22)", {22}, False), ("dec1.conv2 (23)", {23}, False), ("dec1.conv3 (24)", {24}, False),
         ("output.0 (25)", {25}, False), ("all but 0,1,22-25", set(range(26)) - {0, 1, 22, 23, 24, 25}, True),
         ("all but 0-4,19-25", set(range(5, 19)), True))
for name, sub, wg in cases:
    print(f"fp16-rounded weights in {name:44s}: final rel-L2 vs reference {run(sub, wg):.3e}", flush=True)


def bits(ix):
    v = 0
    for i in ix:
        v |= 1 << i
    return v


print("# fp32 weights; ACTIVATIONS rounded to fp16 at the outputs of the named layers only")
for name, ix in (("every layer", list(range(27))), ("enc1 (conv1 = bit 26, conv2, conv3 = x1)", [26, 0, 1]), ("enc2 (2-4)", [2, 3, 4]),
                 ("enc3 .. dec3 (5-18)", list(range(5, 19))), ("dec2 (19-21)", [19, 20, 21]), ("dec1.conv1 (22)", [22]),
                 ("dec1.conv2, conv3, output.0 (23-25)", [23, 24, 25])):
    print(f"fp16-rounded activations after {name:44s}: final rel-L2 vs reference {run(set(), False, bits(ix)):.3e}", flush=True)
