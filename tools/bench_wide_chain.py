"""Dev tool: the two 256-channel chains (csrc/widechain.hip) against the three GEMM launches each replaces, M = 64 x 2048, and the whole
forward with pcd_unet_config(3) v. (1), A/B in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import torch, shapegen_amd
from shapegen_amd import _lib, ops
from shapegen_amd.diffusion import PointCloudDiffusion
from helpers import point_sd
torch.set_grad_enabled(False)
if os.environ.get("WC_LIB"):
    _lib.LIB_PATH = os.environ["WC_LIB"]          # an A/B build of the library (3d-shape-generation_amd/abl_*.so)
lib = _lib.load()
M = 64 * 2048
g = torch.Generator().manual_seed(0)

def ev(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for chain, shapes in ((0, [(256, 256), (256, 256), (512, 256)]), (1, [(256, 512), (256, 256), (128, 256)])):
    ws = [(torch.randn(s, generator=g) / s[1] ** 0.5).half().cuda() for s in shapes]
    bs = [(torch.randn(s[0], generator=g) * 0.1).cuda() for s in shapes]
    x1 = torch.randn(M, 256, generator=g).clamp_min(0).half().cuda()
    x2 = torch.randn(M, 256, generator=g).clamp_min(0).half().cuda()
    packed = torch.empty(int(lib.pcd_pw_wide_packed_bytes(chain)), dtype=torch.uint8, device="cuda")
    wp = (C.c_void_p * 3)(*[t.data_ptr() for t in ws]); bp = (C.c_void_p * 3)(*[t.data_ptr() for t in bs])
    _lib.check(lib.pcd_pw_wide_pack(chain, wp, bp, packed.data_ptr(), _lib.stream_ptr()))
    out = torch.empty(M, shapes[2][0], dtype=torch.float16, device="cuda")
    us = ev(lambda: _lib.check(lib.pcd_pw_wide_chain(chain, x1.data_ptr(), x2.data_ptr() if chain else 0, M, packed.data_ptr(), out.data_ptr(), _lib.stream_ptr())))
    flop = 2.0 * M * sum(s[0] * s[1] for s in shapes)
    # the same three layers as GEMM launches
    def layers():
        a = ops.gemm_f16(x1, ws[0], bs[0], relu=True, a2=x2 if chain else None)
        a = ops.gemm_f16(a, ws[1], bs[1], relu=True)
        return ops.gemm_f16(a, ws[2], bs[2], relu=True)
    try:
        ref = layers()
        us_l = ev(layers)
        err = float((out.float() - ref.float()).norm() / ref.float().norm())
    except Exception as e:
        us_l, err = float("nan"), str(e)[:60]
    print(f"chain {chain}: {us:7.1f} us = {flop / us / 1e6:6.0f} TFLOP/s   three GEMM launches {us_l:7.1f} us   rel diff {err}", flush=True)

if os.environ.get("CHAIN_ONLY"):
    sys.exit(0)
model = PointCloudDiffusion(num_points=2048); model.load_state_dict(point_sd(), strict=True); model = model.to("cuda").eval()
x = torch.randn(64, 2048, 3, device="cuda"); t = torch.rand(64, device="cuda")
for rep in range(2):
    for cfg in (3, 1):
        lib.pcd_unet_config(cfg)
        ms = ev(lambda: model.model(x, t), 20) / 1e3
        if rep: print(f"forward, pcd_unet_config({cfg}): {ms:.3f} ms", flush=True)
lib.pcd_unet_config(3)
