"""Dev tool: run only the dominant GEMM (global_feat.3: M=131072, K=2048, C=4096, fused max) N times, for rocprofv3 --pmc passes.
usage: one_gemm.py <cfg> [reps]   cfg -1: the kernel the U-Net launches (gemm_xw_kernel: fragment-order weights straight from global memory);
cfg >= 0: the LDS-staged kernels with that tile config (pcd_gemm_set_config)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
M, K, C = 64 * 2048, 2048, 4096
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
bias = torch.randn(C, device="cuda", generator=g) * 0.1
if cfg < 0:
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a, w, bias, relu=True)
    r = torch.zeros(M // 2048, C, dtype=torch.float32, device="cuda")
    for _ in range(reps):
        _lib.check(lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), r.data_ptr(), 2048, _lib.stream_ptr()))
else:
    lib.pcd_gemm_set_config(cfg)
    for _ in range(reps):
        r = ops.gemm_f16_colmax(a, w, bias, 2048)
torch.cuda.synchronize()
print("done", float(r.sum()))
