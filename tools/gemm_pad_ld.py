"""Dev tool: does a non-power-of-two leading dimension (row stride) of A / W change the GEMM's L2->LDS feed rate?
argv[1] = stagger bits for the ablation build (0 = normal kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = 64 * 2048
g = torch.Generator(device="cuda").manual_seed(0)
for K, C in [(1024, 2048), (2048, 4096), (512, 512)]:
    line = f"K={K} C={C}:"
    for pad_a, pad_w in [(0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (128, 128), (0, 0)]:
        a = torch.zeros(M, K + pad_a, dtype=torch.float16, device="cuda")
        a[:, :K] = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
        w = torch.zeros(C, K + pad_w, dtype=torch.float16, device="cuda")
        w[:, :K] = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
        bias = torch.zeros(C, device="cuda")
        out = torch.empty(M, C, dtype=torch.float16, device="cuda")
        d = _lib.GemmDesc()
        d.a1, d.lda1, d.k1 = a.data_ptr(), K + pad_a, K
        d.w, d.ldw = w.data_ptr(), K + pad_w
        d.bias, d.relu, d.m, d.c = bias.data_ptr(), 1, M, C
        lib.pcd_gemm_set_config(1000 + bits)
        fn = lambda: _lib.check(lib.pcd_gemm_f16(d, out.data_ptr(), C, _lib.stream_ptr()))
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        line += f"  pad({pad_a},{pad_w}) {e0.elapsed_time(e1) / 5 * 1e3:.0f}us"
    lib.pcd_gemm_set_config(1000)
    print(line, flush=True)
