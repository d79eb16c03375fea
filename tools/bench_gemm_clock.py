"""Dev tool: is the dominant GEMM short of cycles or short of clock?  Runs global_feat.3's shape on gemm_xw_kernel and its mid-tile-request variant back to back, 300 launches each after a 100-launch ramp, with in-kernel stamps (pcd_gemm_wfrag_stamps): per workgroup the shader cycles and
the 100-MHz ticks of its tile walk.  Prints wall time per launch (HIP events), median cycles per K tile and the in-kernel clock = cycles / ticks x 100 MHz."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
M, K, C = 131072, 2048, 4096
a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
bias = torch.randn(C, device="cuda", generator=g) * 0.1
wfrag = torch.empty_like(w)
_lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
d = ops._desc(a, w, bias, relu=True)
out = torch.zeros(M // 2048, C, dtype=torch.float32, device="cuda")
stamps = torch.zeros(256, 2, dtype=torch.int64, device="cuda")
ktiles = (M // 256) * (C // 256) // 256 * (K // 64)
def launch():
    _lib.check(lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
for rnd in range(2):
    for name, cfgs in (("gemm_xw_kernel (one requesting wave per SIMD)", (12, 15)), ("  every wave requests", (12, 14)), ("  mid-tile requests", (13, 14)), ("  k step 0's weights reloaded at the end of the K tile", (12, 15, 33))):
        for c in cfgs: lib.pcd_gemm_set_config(c)
        for _ in range(100): launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): launch()
        e1.record(); torch.cuda.synchronize()
        lib.pcd_gemm_wfrag_stamps(stamps.data_ptr())
        launch()
        torch.cuda.synchronize()
        lib.pcd_gemm_wfrag_stamps(None)
        s = stamps.cpu().double()
        cyc, ticks = s[:, 0].median().item(), s[:, 1].median().item()
        print(f"{name:36s} {e0.elapsed_time(e1) / 300 * 1e3:8.1f} us / launch | {cyc / ktiles:7.0f} cycles per K tile | in-kernel clock {cyc / ticks * 100:6.0f} MHz | stamped launch {ticks / 100:7.1f} us", flush=True)
lib.pcd_gemm_set_config(12); lib.pcd_gemm_set_config(15)
