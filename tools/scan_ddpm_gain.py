"""Dev tool: how large does the state get over `sample2(2, 2048)`'s 1000 steps for a given scaling of the synthetic weights?
(with untamed weights the DDPM loop runs away; a parity fixture over the full horizon needs dynamics that stay O(1-100).)
Runs the fp32 parity mode (equal to the reference to ~1e-6 per forward) with the hashed per-step noise of the G20 capture."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from shapegen_amd import specs  # noqa: E402
from shapegen_amd.diffusion import PointCloudDiffusion  # noqa: E402

torch.set_grad_enabled(False)


class Hashed:
    def __getitem__(self, k):
        return torch.from_numpy(specs.hash_normal(f"g20.z{k}", 2 * 2048 * 3, 0).astype(np.float32).reshape(2, 2048, 3))


torch.manual_seed(11)
xT = torch.randn(2, 2048, 3)
for name, ov in (("gain 1.3, no override", None), ("output.3 x 0.5", {"output.3.weight": 0.5}), ("output.3 x 0.25", {"output.3.weight": 0.25}),
                 ("output.3 x 0.1", {"output.3.weight": 0.1}), ("dec1+output x 0.5", {"output.3.weight": 0.5, "dec1.conv3.weight": 0.5})):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=1.3, overrides=ov).items()}
    m = PointCloudDiffusion(num_points=2048)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    m.model.set_precision("fp32")
    peak = []
    inner = m.model.forward_with_bias

    def fwd(x, tb, stride, out=None, peak=peak, inner=inner):
        peak.append(float(x.abs().max()))
        return inner(x, tb, stride, out=out)

    m.model.forward_with_bias = fwd
    out = m.sample2(2, 2048, x_T=xT.cuda(), noises=Hashed())
    print(f"{name:28s}: |x| max at calls 0/100/250/500/750/900/999 = " + " ".join(f"{peak[k]:.3g}" for k in (0, 100, 250, 500, 750, 900, 999)) +
          f"   final |out| max {float(out.abs().max()):.4g} rms {float(out.pow(2).mean().sqrt()):.4g}", flush=True)
