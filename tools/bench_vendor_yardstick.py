"""Dev tool (yardstick only -- the product never calls a vendor BLAS): torch.matmul (hipBLASLt / rocBLAS behind it) on the point U-Net's
GEMM shapes at cfg2 (M = 64 x 2048 rows, fp16 in / fp16 out, no bias, no ReLU) next to this library's kernels on the same operands
(bias + ReLU + fp16 saturation fused).  Min of 3 x 10 launches each, after a 30-launch ramp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
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

print(f"{'K':>5} {'C':>5} | {'this library':>22} | {'torch.matmul (vendor)':>24}")
for K, C in [(2048, 4096), (1024, 2048), (1024, 1024), (1024, 512), (512, 1024), (512, 512), (512, 256), (256, 512), (256, 256)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    o = torch.empty(M, C, dtype=torch.float16, device="cuda")
    wt = w.t()
    t_v = ev(lambda: torch.matmul(a, wt, out=o))
    if (K, C) == (2048, 4096):
        t_o = ev(lambda: ops.gemm_f16_colmax(a, w, bias, 2048))
        what = "column-max epilogue"
    else:
        t_o = ev(lambda: ops.gemm_f16(a, w, bias, relu=True, out=o))
        what = "store epilogue"
    fl = 2.0 * M * K * C
    print(f"{K:5d} {C:5d} | {t_o:8.1f} us {fl / t_o / 1e6:6.0f} TF/s | {t_v:8.1f} us {fl / t_v / 1e6:6.0f} TF/s   ({what}; the vendor GEMM stores {M * C * 2 / 1e6:.0f} MB)", flush=True)
