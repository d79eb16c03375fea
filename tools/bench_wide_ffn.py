"""Dev tool: the fused FFN of the C = 256 attention blocks (pcd_wide_ffn_f16) alone, B = 64, N = 2048: both request forms (pcd_wide_ffn_config) against the two launches
it replaces (LN2 + Linear + ReLU in the wide-chain kernel, then the K = 1024 GEMM with residual); min of 3 x 20 launches after a 30-launch ramp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
from helpers import sab_sd
torch.set_grad_enabled(False)
lib = _lib.load()
M = 64 * 2048
sd = sab_sd(256)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(M, 256, device="cuda", generator=g) * 1.3).half()
dev = lambda t: t.cuda().contiguous()
w1, b1, w2, b2 = dev(sd["ff.0.weight"].half()), dev(sd["ff.0.bias"].float()), dev(sd["ff.2.weight"].half()), dev(sd["ff.2.bias"].float())
ga, be = dev(sd["ln2.weight"].float()), dev(sd["ln2.bias"].float())
packed = torch.empty(lib.pcd_wide_ffn_packed_bytes(), dtype=torch.uint8, device="cuda")
_lib.check(lib.pcd_wide_ffn_pack(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), ga.data_ptr(), be.data_ptr(), packed.data_ptr(), _lib.stream_ptr()))
lnp = torch.empty(lib.pcd_pw_wide_ln_linear_packed_bytes(4), dtype=torch.uint8, device="cuda")
_lib.check(lib.pcd_pw_wide_ln_linear_pack(w1.data_ptr(), b1.data_ptr(), 4, ga.data_ptr(), be.data_ptr(), lnp.data_ptr(), _lib.stream_ptr()))
y = torch.empty_like(x); y2 = torch.empty_like(x); hid = torch.empty(M, 1024, dtype=torch.float16, device="cuda")
def ev(fn, n=20, reps=3):
    for _ in range(30): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
def two():
    _lib.check(lib.pcd_pw_wide_ln_linear(lnp.data_ptr(), 4, 1, x.data_ptr(), M, hid.data_ptr(), _lib.stream_ptr()))
    ops.gemm_f16_residual(hid, w2, b2, x, out=y2)
fl = 2.0 * M * 256 * 1024 * 2
outs = []
for split in (0, 1):
    lib.pcd_wide_ffn_config(split)
    t = ev(lambda: lib.pcd_wide_ffn_f16(packed.data_ptr(), x.data_ptr(), M, y.data_ptr(), _lib.stream_ptr()))
    outs.append(y.clone())
    print(f"fused FFN, request form {split}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
lib.pcd_wide_ffn_config(1)
for bits, name in ((1, "no image requests in the loop"), (2, "no fragment reads"), (3, "neither"), (4, "no waits / barriers"), (5, "no requests, no barriers"),
                   (8, "no MFMAs"), (9, "no MFMAs, no requests"), (10, "no MFMAs, no fragment reads"), (7, "no requests / reads / barriers (MFMAs only)"),
                   (11, "no MFMAs / requests / reads")):
    lib.pcd_wide_ffn_config(16 + bits)
    t = ev(lambda: lib.pcd_wide_ffn_f16(packed.data_ptr(), x.data_ptr(), M, y.data_ptr(), _lib.stream_ptr()), n=10, reps=2)
    print(f"  timing ablation (outputs wrong): {name:48s} {t:7.1f} us", flush=True)
lib.pcd_wide_ffn_config(16)
lib.pcd_wide_ffn_config(0)
t2 = ev(two)
print(f"two launches (LN + Linear + ReLU, then GEMM + residual): {t2:7.1f} us  {fl / t2 / 1e6:6.0f} TF/s; request forms bitwise equal {torch.equal(outs[0], outs[1])}; "
      f"fused v. two launches max diff {float((outs[0].float() - y2.float()).abs().max()):.3g}", flush=True)
