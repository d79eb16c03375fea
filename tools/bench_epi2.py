"""Dev tool: near-zero-K GEMM = prologue + epilogue only; shows how close the store epilogue is to HBM write speed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
M = 64 * 2048
g = torch.Generator(device="cuda").manual_seed(0)
for K, C in [(64, 2048), (64, 1024), (64, 512), (128, 2048)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    out = torch.empty(M, C, dtype=torch.float16, device="cuda")
    for kind, fn in (("store", lambda: ops.gemm_f16(a, w, bias, relu=True, out=out)), ("colmax", lambda: ops.gemm_f16_colmax(a, w, bias, 2048))):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        print(f"K={K} C={C} {kind}: {best*1e3:.1f} us; output {M*C*2/1e6:.0f} MB -> {M*C*2/best/1e9:.2f} TB/s if store-bound", flush=True)
