"""Dev tool: the attention U-Net's forward at B = 64, N = 2048 with the C <= 128 blocks' tails as one launch each (default) against the four launches
(pcd_sab_tail_config(0)), A/B in one process; and the tail alone per width."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd.networks import SetAttentionBlock
from helpers import sab_sd
torch.set_grad_enabled(False)
lib = _lib.load()
def ev(fn, n=20):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(5)
for C in (64, 128, 256):
    blk = SetAttentionBlock(C, 4); blk.load_state_dict(sab_sd(C), strict=True); blk = blk.to("cuda").eval()
    x = torch.randn(64, 2048, C, generator=g).cuda().half()
    t = {}
    for fused in (1, 3, 0, 1, 3, 0):
        _lib.check(lib.pcd_sab_tail_config(fused))
        t.setdefault(fused, []).append(ev(lambda: blk(x)))
    print(f"SetAttentionBlock C={C}, B=64, N=2048: fused launches {min(t[1]):7.1f} us (every wave requesting its pieces: {min(t[3]):7.1f}) | separate launches {min(t[0]):7.1f} us per block", flush=True)
_lib.check(lib.pcd_sab_tail_config(1))
att = PointCloudDiffusion(num_points=2048, backbone="attention").to("cuda").eval()
x = torch.randn(64, 2048, 3, generator=g).cuda(); tt = torch.rand(64, generator=g).cuda()
for rnd in range(3):
    for fused in (1, 3, 0):
        _lib.check(lib.pcd_sab_tail_config(fused))
        print(f"attention U-Net forward, tails fused={fused}: {ev(lambda: att.model(x, tt)) / 1e3:.3f} ms", flush=True)
_lib.check(lib.pcd_sab_tail_config(1))
