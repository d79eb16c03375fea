"""Dev tool: time the set-attention kernel at BASELINE size (B=64, N=2048, 4 heads): average over 20 launches after
warm-up, for unit-variance and 0.7-sigma qkv, default dispatch and the generic kernel (PCD_ATTN_GENERIC=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
B, N, H = 64, 2048, 4
for generic in ([0, 1, 2] if os.environ.get("PCD_ATTN_BOTH", "1") == "1" else [0]):      # 0 default dispatch, 1 round-1 generic, 2 max-free generic
    _lib.check(lib.pcd_set_attention_config(generic))
    for C in (256, 128, 64):
        for sigma in (1.0, 0.7):
            g = torch.Generator(device="cuda").manual_seed(0)
            qkv = (torch.randn(B * N, 3 * C, device="cuda", generator=g) * sigma).half()
            out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
            def fn():
                _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), 0, 0, _lib.stream_ptr()))
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = 4.0 * B * N * N * C
            print(f"generic={generic} C={C} d={C//H} sigma={sigma}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s "
                  f"({fl/ms/1e9/2500*100:.1f}% of 2.5 PF)", flush=True)
