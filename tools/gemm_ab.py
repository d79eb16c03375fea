"""Dev tool: sweep of the run-time stagger value in the GEMM (pcd_gemm_set_config(1000 + v)) in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
M = 64 * 2048
vals = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
g = torch.Generator(device="cuda").manual_seed(0)
for K, C, kind in [(2048, 4096, "colmax"), (1024, 2048, "f16"), (1024, 1024, "f16"), (512, 1024, "f16"), (512, 512, "f16"), (256, 512, "f16")]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.zeros(C, device="cuda")
    out = torch.empty(M, C, dtype=torch.float16, device="cuda") if kind == "f16" else None
    fn = (lambda: ops.gemm_f16(a, w, bias, relu=True, out=out)) if kind == "f16" else (lambda: ops.gemm_f16_colmax(a, w, bias, 2048))
    res = {v: [] for v in vals}
    for rnd in range(5):
        for v in vals:
            lib.pcd_gemm_set_config(1000 + v)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    lib.pcd_gemm_set_config(1000)
    print(f"K={K} C={C} {kind}: " + " | ".join(f"[{v}] {min(res[v]):.1f}" for v in vals), flush=True)
