"""Dev tool: decoder.12 (Conv3d 32 -> 1, k3 + sigmoid at 32^3, B = 32): 4 x 4 x 8 output blocks (pcd_conv3d_config(8)) against 8 x 8 x 8 (default)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from shapegen_amd import _lib  # noqa: E402

lib = _lib.load()
B = 32
g = torch.Generator().manual_seed(0)
h = torch.randn(B * 32768, 32, generator=g).clamp_min(0).half().cuda()
w32 = (torch.randn(27, 32, generator=g) * 0.1).cuda()
o = [torch.empty(B, 1, 32, 32, 32, device="cuda") for _ in range(3)]


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


wf = torch.empty(lib.pcd_conv3d_last_packed_bytes(), dtype=torch.uint8, device="cuda")
_lib.check(lib.pcd_conv3d_last_pack(w32.data_ptr(), wf.data_ptr(), _lib.stream_ptr()))
o.append(torch.empty_like(o[0]))
o.append(torch.empty_like(o[0]))
for rep in range(3):
    t = []
    for k, cfg in enumerate((9, 17, 25, 16385, 1)):
        _lib.check(lib.pcd_conv3d_config(cfg))
        t.append(ev(lambda: _lib.check(lib.pcd_conv3d_last_sigmoid_packed(h.data_ptr(), B, 32, 32, 32, 32, w32.data_ptr(), wf.data_ptr(), 0.05, o[k].data_ptr(),
                                                                          _lib.stream_ptr()))))
    print(f"4x4x8 blocks {t[0]:6.1f} us | 8x8x8 blocks, VALU {t[1]:6.1f} us | 8x8x8 blocks, one MFMA per tap {t[2]:6.1f} us | per-tap partial products, a wave per slice {t[3]:6.1f} us | a wave per four slices (default) {t[4]:6.1f} us | "
          f"max diff {float((o[0] - o[1]).abs().max()):.1e} {float((o[0] - o[2]).abs().max()):.1e} {float((o[0] - o[3]).abs().max()):.1e} {float((o[0] - o[4]).abs().max()):.1e}", flush=True)
for cfg, name in ((1 + 128 + 16384, "loads only"), (1 + 256 + 16384, "no loads")):
    _lib.check(lib.pcd_conv3d_config(cfg))
    t = ev(lambda: _lib.check(lib.pcd_conv3d_last_sigmoid_packed(h.data_ptr(), B, 32, 32, 32, 32, w32.data_ptr(), wf.data_ptr(), 0.05, o[3].data_ptr(), _lib.stream_ptr())))
    print(f"per-tap partial products, timing ablation (outputs wrong): {name:12s} {t:6.1f} us", flush=True)
_lib.check(lib.pcd_conv3d_config(1))
