"""Dev tool (yardstick): a few torch.matmul launches on the K = 1024 / 2048 shapes, for rocprofv3 --kernel-trace (kernel names of the vendor GEMMs)."""
import torch
M = 64 * 2048
for K, C in [(2048, 4096), (1024, 2048), (1024, 1024), (512, 512)]:
    a = torch.randn(M, K, device="cuda").clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda") / K ** 0.5).half()
    o = torch.empty(M, C, dtype=torch.float16, device="cuda")
    for _ in range(5): torch.matmul(a, w.t(), out=o)
torch.cuda.synchronize()
