"""Dev tool: wall-clock time of VAE3DLarge.decode / .encode at B = 32 (HIP events around 10 back-to-back calls after warm-up),
next to the FLOP counts of SURVEY section 8(d) (40.20 / 24.48 GFLOP per sample).  The per-layer tables in profiles/ come from
rocprofv3 --kernel-trace, which serialises the launches: their sums are 10-15 % above these times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import shapegen_amd
from shapegen_amd import specs
from shapegen_amd.vae import VAE3DLarge
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
vae = VAE3DLarge()
sd = specs.synth_state_dict(specs.vae3d_large_spec(), seed=0, gain=1.3)
vae.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
vae = vae.to("cuda").eval()
z = torch.randn(B, 256, device="cuda")
x = (torch.rand(B, 1, 32, 32, 32, device="cuda") > 0.9).float()
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
from shapegen_amd import _lib
cfg = int(os.environ.get("PCD_CONV3D_CONFIG", "1"))
_lib.check(_lib.load().pcd_conv3d_config(cfg))
for rep in range(3):
    td, te = timed(lambda: vae.decode(z)), timed(lambda: vae.encode(x))
    print(f"B={B}: decode {td:7.1f} us = {40.20e9 * B / td / 1e6:5.0f} TFLOP/s = {40.20e9 * B / td / 1e6 / 25:4.1f} % of 2.5 PF | "
          f"encode {te:7.1f} us = {24.48e9 * B / te / 1e6:5.0f} TFLOP/s = {24.48e9 * B / te / 1e6 / 25:4.1f} %", flush=True)
