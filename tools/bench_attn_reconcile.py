"""Dev tool: which set-attention kernel does the default dispatch launch per head width, and how fast is each
candidate under the SAME conditions -- 200 launches of ramp, then 100 timed, unit-variance q / k / v at B = 64, N = 2048, 4 heads --
in two visiting orders (the chip is power limited: what ran before a leg changes its clocks).  Prints the name the library reports
(`pcd_set_attention_last_kernel`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import shapegen_amd  # noqa: E402,F401
from shapegen_amd import _lib  # noqa: E402

lib = _lib.load()
B, N, H = 64, 2048, 4
bufs = {}
for C in (256, 128, 64):
    g = torch.Generator(device="cpu").manual_seed(7)
    bufs[C] = (torch.randn(B * N, 3 * C, generator=g).to("cuda", torch.float16), torch.empty(B * N, C, dtype=torch.float16, device="cuda"))


def leg(C, generic, warm=200, count=100):
    qkv, out = bufs[C]
    _lib.check(lib.pcd_set_attention_config(generic))

    def fn():
        _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), 0, 0, _lib.stream_ptr()))
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(count):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / count
    name = lib.pcd_set_attention_last_kernel().decode()
    tf = 4.0 * B * N * N * C / ms / 1e9
    print(f"C={C:3d} d={C // H:2d} config={generic} kernel={name:30s} {ms * 1e3:7.1f} us  {tf:6.0f} TFLOP/s ({tf / 25:.1f} %)", flush=True)


order = [(C, gcfg) for C in (256, 128, 64) for gcfg in (0, 1, 2)]
print("# order A: d = 64, 32, 16; default, round-1 generic, max-free generic")
for C, gcfg in order:
    leg(C, gcfg)
print("# order B: reversed")
for C, gcfg in reversed(order):
    leg(C, gcfg)
print("# order C: each default-dispatch leg after 2 s of idle")
import time  # noqa: E402
for C in (256, 128, 64):
    time.sleep(2.0)
    leg(C, 0)
_lib.check(lib.pcd_set_attention_config(0))
