"""Dev tool: where the store epilogue's time goes.  gemm_xs_kernel's ablation instance (pcd_gemm_set_config(16 + bits); outputs are WRONG while set):
bit 0 no direct stores (dummy 4-byte loads keep the vmcnt counts), bit 1 no drip stores (likewise), bit 2 no LDS parking, bit 3 no epilogue arithmetic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
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
M = 131072
names = {0: "product kernel", 1: "no direct stores", 2: "no drip stores", 3: "no global stores (arithmetic + LDS parking + read-back)", 7: "epilogue arithmetic only",
         11: "K loop + LDS parking traffic of garbage", 15: "K loop only"}
for K, C in [(1024, 2048), (1024, 1024), (512, 512)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a, w, bias, relu=True)
    out = torch.empty(M, C, dtype=torch.float16, device="cuda")
    t_p = ev(lambda: lib.pcd_gemm_f16(d, out.data_ptr(), C, _lib.stream_ptr()))
    print(f"K={K} C={C}: gemm_xp_kernel {t_p:8.1f} us", flush=True)
    for abl in (0, 1, 2, 3, 7, 15):
        lib.pcd_gemm_set_config(16 + abl)
        t = ev(lambda: lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
        print(f"    ablation {abl:2d} ({names[abl]}): {t:8.1f} us", flush=True)
    lib.pcd_gemm_set_config(16)
    del a, w, out
