"""Dev tool: run only the widest store GEMM (global_feat.0: M=131072, K=1024, C=2048, fp16 store epilogue) N times with a given
pcd_gemm_set_config value (7: tile-boundary kernel, 5: generic), for rocprofv3 --pmc passes.  usage: one_gemm_store.py <cfg> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
M, K, C = 64 * 2048, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
bias = torch.randn(C, device="cuda", generator=g) * 0.1
out = torch.empty(M, C, dtype=torch.float16, device="cuda")
lib.pcd_gemm_set_config(cfg)
for _ in range(reps):
    ops.gemm_f16(a, w, bias, relu=True, out=out)
torch.cuda.synchronize()
print("done", float(out.float().sum()))
