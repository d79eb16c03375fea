"""Dev tool: the column-max GEMM with fragment-order weights straight from global memory (gemm_xw_kernel) against gemm_xp_kernel on the global_feat.3 shape
and two smaller ones: bitwise check, then min of 3 x 10 launches after a 30-launch ramp, A/B in one process."""
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
for M, K, C in [(131072, 2048, 4096), (131072, 1024, 2048), (131072, 512, 512)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a, w, bias, relu=True)
    out = torch.zeros(M // 2048, C, dtype=torch.float32, device="cuda")
    def xw():
        out.zero_()
        _lib.check(lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
    xw()
    ref = ops.gemm_f16_colmax(a, w, bias, 2048)
    same = torch.equal(out, ref)
    t_w = ev(lambda: lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
    lib.pcd_gemm_set_config(13)        # activation pieces of the next K tile requested behind k step 0's MFMAs
    xw()
    same_mid = torch.equal(out, ref)
    t_m = ev(lambda: lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
    lib.pcd_gemm_set_config(33)        # diagnostic: k step 0's weights reloaded late
    xw()
    same_late = torch.equal(out, ref)
    lib.pcd_gemm_set_config(12)
    print("      (late-weights diagnostic bitwise equal:", same_late, ")", flush=True)
    lib.pcd_gemm_set_config(14)        # one wave per SIMD requests the activation pieces (8 each), alternating groups per K tile
    xw()
    same_split = torch.equal(out, ref)
    t_s = ev(lambda: lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
    lib.pcd_gemm_set_config(12)
    t_p = ev(lambda: lib.pcd_gemm_f16_colmax(d, out.data_ptr(), 2048, _lib.stream_ptr()))
    fl = 2.0 * M * K * C
    print(f"M={M} K={K} C={C}: weights from global {t_w:8.1f} us {fl / t_w / 1e6:6.0f} TF/s | mid-tile requests {t_m:8.1f} us {fl / t_m / 1e6:6.0f} TF/s (equal {same_mid}) | split requests {t_s:8.1f} us {fl / t_s / 1e6:6.0f} TF/s (equal {same_split}) | gemm_xp_kernel {t_p:8.1f} us {fl / t_p / 1e6:6.0f} TF/s | bitwise equal {same}", flush=True)
