"""Dev tool: the cross-tile-prefetching 256x256 store kernel (gemm_xp_kernel) against the generic kernel on the point U-Net's store shapes,
A/B in one process (pcd_gemm_set_config(6 / 5)), outputs compared bitwise; then the whole forward (MI355X only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
M = 64 * 2048
g = torch.Generator(device="cuda").manual_seed(0)

def ev(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for K, C in [(1024, 2048), (1024, 1024), (1024, 512), (512, 1024), (512, 512), (512, 256), (256, 512), (256, 256), (128, 256)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    out = {}
    t = {}
    for rnd in range(2):
        for cfg in (6, 7, 5):
            lib.pcd_gemm_set_config(cfg)
            o = torch.empty(M, C, dtype=torch.float16, device="cuda")
            fn = lambda: ops.gemm_f16(a, w, bias, relu=True, out=o)
            us = ev(fn)
            t[cfg] = min(t.get(cfg, 1e9), us)
            out[cfg] = o
    same = torch.equal(out[5], out[6]) and torch.equal(out[5], out[7])
    print(f"K={K:5d} C={C:5d}: xp2 {t[6]:7.1f} us {2.0*M*K*C/t[6]/1e6:6.0f} TF | xp1 {t[7]:7.1f} us {2.0*M*K*C/t[7]/1e6:6.0f} TF | generic {t[5]:7.1f} us {2.0*M*K*C/t[5]/1e6:6.0f} TF | bitwise equal {same}", flush=True)

# the column-max GEMM (global_feat.3): same kernel, first K tile of the next tile requested during the last one
a = torch.randn(M, 2048, device="cuda", generator=g).clamp_min(0).half()
w = (torch.randn(4096, 2048, device="cuda", generator=g) / 2048 ** 0.5).half()
bias = torch.randn(4096, device="cuda", generator=g) * 0.1
t, out = {}, {}
for rnd in range(2):
    for cfg in (7, 5):
        lib.pcd_gemm_set_config(cfg)
        us = ev(lambda: ops.gemm_f16_colmax(a, w, bias, 2048))
        t[cfg] = min(t.get(cfg, 1e9), us)
        out[cfg] = ops.gemm_f16_colmax(a, w, bias, 2048).clone()
print(f"colmax K=2048 C=4096: xp {t[7]:7.1f} us {2.0*M*2048*4096/t[7]/1e6:6.0f} TF | generic {t[5]:7.1f} us {2.0*M*2048*4096/t[5]/1e6:6.0f} TF | bitwise equal {torch.equal(out[5], out[7])}", flush=True)

from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); tt = torch.rand(64, device="cuda")
res = {}
for rep in range(3):
    for cfg in (6, 7, 5):
        lib.pcd_gemm_set_config(cfg)
        ms = ev(lambda: model.model(x, tt), 20) / 1e3
        res[cfg] = model.model(x, tt).clone()
        if rep: print(f"forward, {'xp 2' if cfg == 6 else ('xp 1' if cfg == 7 else 'generic')}: {ms:.3f} ms", flush=True)
print("forward outputs bitwise equal:", torch.equal(res[5], res[6]) and torch.equal(res[5], res[7]))
lib.pcd_gemm_set_config(7)
