"""Dev tool: fp16 product path against the fp32 parity mode over `sample2(2, 2048)`'s 1000 steps, for several scalings of the synthetic
weights (the fp32 mode equals the reference to ~1e-6 per forward, so this predicts the parity of a reference capture with those weights)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from helpers import rel_l2  # noqa: E402
from shapegen_amd import specs  # noqa: E402
from shapegen_amd.diffusion import PointCloudDiffusion  # noqa: E402

torch.set_grad_enabled(False)


class Hashed:
    def __getitem__(self, k):
        return torch.from_numpy(specs.hash_normal(f"g20.z{k}", 2 * 2048 * 3, 0).astype(np.float32).reshape(2, 2048, 3))


torch.manual_seed(11)
xT = torch.randn(2, 2048, 3)
cases = (("gain 1.3, no override", 1.3, None), ("output.3 x 0.5", 1.3, {"output.3.weight": 0.5}), ("output.3 x 0.25", 1.3, {"output.3.weight": 0.25}),
         ("output.3 x 0.1", 1.3, {"output.3.weight": 0.1}), ("gain 1.0", 1.0, None), ("gain 0.8", 0.8, None))
for name, gain, ov in cases:
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=gain, overrides=ov).items()}
    res = {}
    for prec in ("fp32", "fp16"):
        m = PointCloudDiffusion(num_points=2048)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda").eval()
        m.model.set_precision(prec)
        st = []
        inner = m.model.forward_with_bias

        def fwd(x, tb, stride, out=None, st=st, inner=inner):
            st.append(x.detach().clone())
            return inner(x, tb, stride, out=out)

        m.model.forward_with_bias = fwd
        out = m.sample2(2, 2048, x_T=xT.cuda(), noises=Hashed())
        res[prec] = (st, out)
    s32, o32 = res["fp32"]
    s16, o16 = res["fp16"]
    print(f"{name:24s}: final |x| max {float(o32.abs().max()):.3g} rms {float(o32.pow(2).mean().sqrt()):.3g}; fp16 vs fp32 rel-L2 at calls 100/250/500/750/999/final = " +
          " ".join(f"{rel_l2(s16[k].cpu(), s32[k].cpu()):.2e}" for k in (100, 250, 500, 750, 999)) + f" {rel_l2(o16.cpu(), o32.cpu()):.2e}", flush=True)
