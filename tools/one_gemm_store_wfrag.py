"""Dev tool: run only the widest store GEMM (global_feat.0: M=131072, K=1024, C=2048) N times on the kernel the U-Net launches for it
(gemm_xs_kernel through pcd_gemm_f16_wfrag), for rocprofv3 --pmc passes.  usage: one_gemm_store_wfrag.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M, K, C = 64 * 2048, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
bias = torch.randn(C, device="cuda", generator=g) * 0.1
wfrag = torch.empty_like(w)
_lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
d = ops._desc(a, w, bias, relu=True)
out = torch.empty(M, C, dtype=torch.float16, device="cuda")
for _ in range(reps):
    _lib.check(lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", float(out[:4].float().sum()))
