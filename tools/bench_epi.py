"""Dev tool: store epilogue vs no-store (colmax) epilogue time on mid-size layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
M = 64 * 2048
g = torch.Generator(device="cuda").manual_seed(0)
for K, C in [(1024, 2048), (1024, 1024), (512, 1024), (512, 512), (256, 256)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    out = torch.empty(M, C, dtype=torch.float16, device="cuda")
    res = {}
    for stg in (0, 100, 200, 400):
        lib.pcd_gemm_set_config(1000 + stg)
        fn = lambda: ops.gemm_f16(a, w, bias, relu=True, out=out)
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        print(f"   stagger={stg}: {best*1e3:.1f} us")
    lib.pcd_gemm_set_config(1000)
    for kind, fn in (("store", lambda: ops.gemm_f16(a, w, bias, relu=True, out=out)), ("colmax", lambda: ops.gemm_f16_colmax(a, w, bias, 2048))):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        res[kind] = best
    print(f"K={K} C={C}: store {res['store']*1e3:.1f} us ({2.0*M*K*C/res['store']/1e9:.0f} TF)  no-store {res['colmax']*1e3:.1f} us ({2.0*M*K*C/res['colmax']/1e9:.0f} TF)  write {M*C*2/1e6:.0f} MB", flush=True)
