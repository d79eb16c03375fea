"""Dev tool: gemm_xp_kernel's store epilogue with and without the non-temporal hint on its 16-byte stores (pcd_gemm_set_config(11) / (10)),
per store-layer shape of the point U-Net at cfg2 and on the whole forward.  Min of 3 x 10 launches after a 30-launch ramp; outputs compared bitwise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
M = 64 * 2048
g = torch.Generator(device="cuda").manual_seed(0)

def ev(fn, n=10, reps=3):
    for _ in range(30): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

for K, C in [(1024, 2048), (1024, 1024), (1024, 512), (512, 1024), (512, 512), (512, 256), (256, 512), (256, 256)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    o = [torch.empty(M, C, dtype=torch.float16, device="cuda") for _ in range(2)]
    t = []
    for nt in (0, 1, 0, 1):
        _lib.check(lib.pcd_gemm_set_config(10 + nt))
        t.append(ev(lambda: ops.gemm_f16(a, w, bias, relu=True, out=o[nt])))
    fl = 2.0 * M * K * C
    print(f"K={K:5d} C={C:5d}: plain {min(t[0], t[2]):7.1f} us {fl / min(t[0], t[2]) / 1e6:5.0f} TF | nt {min(t[1], t[3]):7.1f} us {fl / min(t[1], t[3]) / 1e6:5.0f} TF | bitwise equal {torch.equal(o[0], o[1])}", flush=True)

# whole forward
from shapegen_amd.diffusion import PointCloudDiffusion
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import point_sd
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); tt = torch.rand(64, device="cuda")
net = model.model
for rnd in range(3):
    for nt in (0, 1):
        _lib.check(lib.pcd_gemm_set_config(10 + nt))
        print(f"forward, nt={nt}: {ev(lambda: net(x, tt), n=20, reps=1) / 1e3:.3f} ms", flush=True)
_lib.check(lib.pcd_gemm_set_config(10))
