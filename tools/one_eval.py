"""Dev tool: the evaluation of test_point_ddpm.py:85-92 at BASELINE size -- 64 pairs of 2048-point clouds, Chamfer +
Sinkhorn EMD + voxel BCE -- through metrics.pair_metrics (one enqueue), 5 times (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import metrics as M
g = torch.Generator().manual_seed(0)
a = (torch.rand(64, 2048, 3, generator=g) * 2 - 1).cuda()
b = (a + 0.05 * torch.randn(64, 2048, 3, generator=g).cuda()).contiguous()
for approx in (True,):
    for _ in range(2):
        M.pair_metrics(a, b, approx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        rows = M.pair_metrics(a, b, approx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"pair_metrics(64 x 2048 x 2048, sinkhorn={approx}): {dt * 1e3:.2f} ms per evaluation; mean rows {rows.mean(0).tolist()}")
