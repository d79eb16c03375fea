"""Dev tool: time the tuning variants of the d = 64 set-attention kernel (pcd_set_attention_config(v))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
B, N, H, C = 64, 2048, 4, 256
variants = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 5, 0]
ref = None
for sigma in (1.0, 0.7):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B * N, 3 * C, device="cuda", generator=g) * sigma).half()
    out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
    for v in variants:
        _lib.check(lib.pcd_set_attention_config(v))
        def fn():
            _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), 0, 0, _lib.stream_ptr()))
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        if v == 1:
            ref = out.float().clone()
        err = float((out.float() - ref).norm() / ref.norm()) if ref is not None else -1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 4.0 * B * N * N * C
        print(f"variant={v} sigma={sigma}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s ({fl/ms/1e9/2500*100:.1f}%)  rel diff vs generic {err:.1e}", flush=True)
_lib.check(lib.pcd_set_attention_config(0))
