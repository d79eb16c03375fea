"""Dev tool: run only the set-attention kernel at BASELINE size (B=64, N=2048, C=256, H=4) for PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, N, H = 64, 2048, 4
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * N, 3 * C, device="cuda", generator=g) * 0.7).half()
ws = torch.empty(max(16, lib.pcd_set_attention_workspace_bytes(B, N, C)), dtype=torch.uint8, device="cuda")
out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
for _ in range(6):
    _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
torch.cuda.synchronize()
print("done")
