"""Dev tool for rocprofv3: 30 forwards of the attention U-Net at B = 64, N = 2048 (tails fused unless argv[1] == 0)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import PointCloudDiffusion
torch.set_grad_enabled(False)
lib = _lib.load()
_lib.check(lib.pcd_sab_tail_config(int(sys.argv[1]) if len(sys.argv) > 1 else 1))
att = PointCloudDiffusion(num_points=2048, backbone="attention").to("cuda").eval()
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 2048, 3, generator=g).cuda(); tt = torch.rand(64, generator=g).cuda()
for _ in range(30): att.model(x, tt)
torch.cuda.synchronize()
