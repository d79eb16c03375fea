"""Dev tool: VAE3DLarge.encode at B = 32 with the residual blocks' projection shortcuts inside conv2's launch (default) against the
separate pointwise launch + residual read (set_fuse_shortcut(False)), A/B in one process, HIP events around 20 back-to-back calls."""
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
x = (torch.rand(B, 1, 32, 32, 32, device="cuda") > 0.9).float()
def timed(fn, n=20):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    for fuse in (True, False):
        vae.set_fuse_shortcut(fuse)
        te = timed(lambda: vae.encode(x))
        print(f"B={B} fused={int(fuse)}: encode {te:7.1f} us = {24.48e9 * B / te / 1e6:5.0f} TFLOP/s = {24.48e9 * B / te / 1e6 / 25:4.1f} % of 2.5 PF", flush=True)
vae.set_fuse_shortcut(True)
# decoder.6 through the LDS-halo transposed-convolution kernel (default) against the eight implicit-GEMM class launches
from shapegen_amd import _lib
lib = _lib.load()
z = torch.randn(B, 256, device="cuda")
for rep in range(3):
    for halo in (1, 0):
        _lib.check(lib.pcd_vae_config(halo))
        td, te = timed(lambda: vae.decode(z)), timed(lambda: vae.encode(x))
        print(f"B={B} LDS kernels for decoder.6 / encoder.3 = {halo}: decode {td:7.1f} us = {40.20e9 * B / td / 1e6 / 25:4.1f} % | encode {te:7.1f} us = {24.48e9 * B / te / 1e6 / 25:4.1f} % of 2.5 PF", flush=True)
_lib.check(lib.pcd_vae_config(1))
