"""Dev tool: time the set-attention kernel (and the whole SetAttentionBlock) at BASELINE size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
lib = _lib.load()
B, N, H = 64, 2048, 4
for C in (256, 128, 64):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B * N, 3 * C, device="cuda", generator=g) * 0.7).half()
    ws = torch.empty(max(16, lib.pcd_set_attention_workspace_bytes(B, N, C)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
    def fn():
        _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = 4.0 * B * N * N * C
    print(f"C={C} d={C//H}: {best*1e3:.1f} us  {fl/best/1e9:.0f} TFLOP/s ({fl/best/1e9/2500*100:.1f}% of 2.5 PF)", flush=True)
