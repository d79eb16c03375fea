"""Dev tool (round 5): "one requesting wave per SIMD" -- the LDS-DMA pieces of a stage requested by waves 0-3 (even stages) / 4-7 (odd stages), 8 pieces each, instead of 4
pieces by every wave -- A/B per kernel family and on the two U-Net forwards, one process.  Switches: pcd_gemm_set_config(14 | 15) (gemm_xw / gemm_xs),
pcd_pw_wide_config (wide chains, LN + Linear), pcd_wide_ffn_config (fused FFN).  Outputs must be bitwise equal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import numpy as np
import torch, shapegen_amd
from shapegen_amd import _lib, ops, specs
from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
torch.set_grad_enabled(False)
lib = _lib.load()
M = 64 * 2048
g = torch.Generator().manual_seed(0)

def ev(fn, n=20, reps=3, ramp=30):
    for _ in range(ramp): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

def both(setter, fn, result):
    t, outs = [], []
    for on in (0, 1):
        setter(on)
        t.append(ev(fn)); outs.append(result().clone())
    setter(1)
    return t[0], t[1], torch.equal(outs[0], outs[1])

# ---- the two wide chains of the point U-Net
for chain, shapes in ((0, [(256, 256), (256, 256), (512, 256)]), (1, [(256, 512), (256, 256), (128, 256)])):
    ws = [(torch.randn(s, generator=g) / s[1] ** 0.5).half().cuda() for s in shapes]
    bs = [(torch.randn(s[0], generator=g) * 0.1).cuda() for s in shapes]
    x1 = torch.randn(M, 256, generator=g).clamp_min(0).half().cuda()
    x2 = torch.randn(M, 256, generator=g).clamp_min(0).half().cuda()
    packed = torch.empty(int(lib.pcd_pw_wide_packed_bytes(chain)), dtype=torch.uint8, device="cuda")
    wp = (C.c_void_p * 3)(*[t.data_ptr() for t in ws]); bp = (C.c_void_p * 3)(*[t.data_ptr() for t in bs])
    _lib.check(lib.pcd_pw_wide_pack(chain, wp, bp, packed.data_ptr(), _lib.stream_ptr()))
    out = torch.empty(M, shapes[2][0], dtype=torch.float16, device="cuda")
    a, b, same = both(lib.pcd_pw_wide_config, lambda: lib.pcd_pw_wide_chain(chain, x1.data_ptr(), x2.data_ptr() if chain else 0, M, packed.data_ptr(), out.data_ptr(), _lib.stream_ptr()), lambda: out)
    print(f"wide chain {chain}: every wave requests {a:7.1f} us | one wave per SIMD {b:7.1f} us | bitwise equal {same}", flush=True)

# ---- the dominant GEMM and the widest store GEMM
for K, Cc, colmax in ((2048, 4096, True), (1024, 2048, False), (512, 512, False)):
    a_ = torch.randn(M, K, generator=g).clamp_min(0).half().cuda()
    w = (torch.randn(Cc, K, generator=g) / K ** 0.5).half().cuda()
    bias = (torch.randn(Cc, generator=g) * 0.1).cuda()
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, Cc, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a_, w, bias, relu=True)
    if colmax:
        out = torch.zeros(M // 2048, Cc, dtype=torch.float32, device="cuda")
        def fn():
            out.zero_()
            lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr())
    else:
        out = torch.empty(M, Cc, dtype=torch.float16, device="cuda")
        fn = lambda: lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), Cc, _lib.stream_ptr())
    a, b, same = both(lambda on: lib.pcd_gemm_set_config(14 + on), fn, lambda: out)
    print(f"GEMM K = {K}, C = {Cc} ({'column max' if colmax else 'store'}): every wave requests {a:7.1f} us | one wave per SIMD {b:7.1f} us | bitwise equal {same}", flush=True)
    del a_, w, out

# ---- the forwards
def all_switches(on):
    lib.pcd_gemm_set_config(14 + on); lib.pcd_pw_wide_config(on); lib.pcd_wide_ffn_config(on)      # (the LN + Linear launches keep "every wave requests" at 1)
pm = PointCloudDiffusion(num_points=2048); pm.load_state_dict(point_sd(), strict=True); pm = pm.to("cuda").eval()
am = PointCloudDiffusion(num_points=2048, backbone="attention")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in specs.synth_state_dict(specs.unet_attention_spec(), seed=0, gain=1.0).items()}
am.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=True); am = am.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); t = torch.rand(64, device="cuda")
for name, m in (("point U-Net", pm), ("attention U-Net", am)):
    res = {}
    for rnd in range(3):
        for on in (1, 0):
            all_switches(on)
            ms = ev(lambda: m.model(x, t), n=20, reps=1, ramp=5) / 1e3
            res[on] = m.model(x, t).clone()
            print(f"{name} forward, {'one requesting wave per SIMD' if on else 'every wave requests'}: {ms:.3f} ms", flush=True)
    print(f"{name}: outputs bitwise equal {torch.equal(res[0], res[1])}", flush=True)
all_switches(1)
