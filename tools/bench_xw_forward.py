"""Dev tool: the point U-Net forward at cfg2 (B = 64, N = 2048) with global_feat.3 on gemm_xw_kernel (fragment-order weights straight from global memory,
default) against gemm_xp_kernel (pcd_gemm_set_config 9 / 8), A/B in one process: 4 rounds x 20 forwards after a 10-forward ramp; outputs bitwise equal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
torch.set_grad_enabled(False)
lib = _lib.load()
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); tt = torch.rand(64, device="cuda")
def ev(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {8: [], 9: []}
for rep in range(4):
    for cfg in (9, 8):
        lib.pcd_gemm_set_config(cfg)
        res[cfg].append(ev(lambda: model.model(x, tt)))
        out = model.model(x, tt).clone()
        if cfg == 9: ref = out
        else: assert torch.equal(out, ref)
lib.pcd_gemm_set_config(9)
print("global_feat.3 on gemm_xw_kernel (weights from global): " + "  ".join(f"{v:.3f}" for v in res[9]) + " ms / forward")
print("global_feat.3 on gemm_xp_kernel (LDS-staged)         : " + "  ".join(f"{v:.3f}" for v in res[8]) + " ms / forward")
print("outputs bitwise equal")
