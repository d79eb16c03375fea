"""Dev tool: sustained rate of the set-attention kernel (B=64, N=2048, C=256): 10 chunks of 100 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
B, N, H, C = 64, 2048, 4, 256
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).half()
out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
for generic in (0, 1):
    _lib.check(lib.pcd_set_attention_config(generic))
    res = []
    for chunk in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), 0, 0, _lib.stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        res.append(4.0 * B * N * N * C / (e0.elapsed_time(e1) / 100) / 1e9)
    print(f"generic={generic}: TFLOP/s per 100-launch chunk: " + " ".join(f"{r:.0f}" for r in res), flush=True)
