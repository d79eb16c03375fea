"""Dev tool: encoder.0 (Conv3d 1 -> 32, k3 + BN + ReLU at 32^3, B = 32): 8 x 8 x 8 tiles (default) against 4 x 4 x 8 tiles (pcd_conv3d_config(+ 512)); same bits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from shapegen_amd import _lib  # noqa: E402

lib = _lib.load()
B = 32
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 1, 32, 32, 32, generator=g) > 0.9).float().cuda()
w = (torch.randn(32, 27, generator=g) * 0.3).cuda()
b = (torch.randn(32, generator=g) * 0.1).cuda()
o = [torch.empty(B * 32768, 32, dtype=torch.float16, device="cuda") for _ in range(2)]


def ev(fn, n=50):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(3):
    t = []
    for k, cfg in enumerate((513, 1)):
        _lib.check(lib.pcd_conv3d_config(cfg))
        t.append(ev(lambda: _lib.check(lib.pcd_conv3d_first(x.data_ptr(), B, 32, 32, 32, 1, w.data_ptr(), b.data_ptr(), 32, o[k].data_ptr(), _lib.stream_ptr()))))
    print(f"4x4x8 tiles {t[0]:6.1f} us | 8x8x8 tiles (default) {t[1]:6.1f} us | same bits {torch.equal(o[0], o[1])}", flush=True)
_lib.check(lib.pcd_conv3d_config(1))
