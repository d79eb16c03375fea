"""Dev tool: the Chamfer part of pcd_pair_metrics (64 pairs of 2048 x 2048 points, no Sinkhorn): one pass that evaluates every distance once
against the two one-direction passes (pcd_pair_metrics_config)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
P, N = 64, 2048
g = torch.Generator().manual_seed(0)
a = (torch.rand(P, N, 3, generator=g) * 2 - 1).cuda(); b = (torch.rand(P, N, 3, generator=g) * 2 - 1).cuda()
na = torch.full((P,), N, dtype=torch.int32, device="cuda"); nb = na.clone()
rows = torch.empty(P, 3, device="cuda")
need = int(lib.pcd_pair_metrics_workspace_bytes(P, N, N)); ws = torch.empty(need, dtype=torch.uint8, device="cuda")
def run():
    _lib.check(lib.pcd_pair_metrics(a.data_ptr(), na.data_ptr(), N, b.data_ptr(), nb.data_ptr(), N, P, 0, 1e-2, 1e-5, 100, 0, 0, rows.data_ptr(),
                                    ws.data_ptr(), need, _lib.stream_ptr()))
for mode in (1, 0, 1, 0):
    lib.pcd_pair_metrics_config(mode)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"chamfer two_pass={mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per evaluation (normalize + Chamfer + voxel BCE, 64 pairs)  cd[0]={float(rows[0,0]):.6f}", flush=True)
lib.pcd_pair_metrics_config(0)
