"""Dev tool: the store GEMMs of the point U-Net on gemm_xs_kernel (weights from global memory, output dripped through LDS during the next tile's K loop)
against gemm_xp_kernel (both operands through LDS, store burst behind two prefetched K tiles), same operands: bitwise check, then min of 3 x 10 launches after a
30-launch ramp, A/B in one process.  `--forward`: the whole U-Net forward at cfg2 with the kernel on / off (4 rounds x 20)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops
torch.set_grad_enabled(False)
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
def ev(fn, n=10, reps=3):
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
M = 131072
tot_s = tot_p = 0.0
# (K, C, launches per U-Net forward): global_feat.0; dec4.conv1-2; dec4.conv3 + dec3.conv1; enc4.conv3; enc4.conv1-2 + dec3.conv2; dec3.conv3
for K, C, n in [(1024, 2048, 1), (1024, 1024, 2), (1024, 512, 2), (512, 1024, 1), (512, 512, 3), (512, 256, 1)]:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), K, K, C, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a, w, bias, relu=True)
    out = torch.empty(M, C, dtype=torch.float16, device="cuda")
    ref = ops.gemm_f16(a, w, bias, relu=True)
    _lib.check(lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
    same = torch.equal(out, ref)
    t_s = ev(lambda: lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
    lib.pcd_gemm_set_config(13)        # the next K tile's activation pieces requested behind k step 0's MFMAs instead of at the top of the K tile
    out.zero_()
    _lib.check(lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
    same_mid = torch.equal(out, ref)
    t_m = ev(lambda: lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), C, _lib.stream_ptr()))
    lib.pcd_gemm_set_config(12)
    t_p = ev(lambda: lib.pcd_gemm_f16(d, out.data_ptr(), C, _lib.stream_ptr()))
    tot_s += n * t_s; tot_p += n * t_p
    fl = 2.0 * M * K * C
    print(f"K={K:5d} C={C:5d}: gemm_xs_kernel {t_s:8.1f} us {fl / t_s / 1e6:6.0f} TF/s | mid-tile requests {t_m:8.1f} us {fl / t_m / 1e6:6.0f} TF/s (equal {same_mid}) | gemm_xp_kernel {t_p:8.1f} us {fl / t_p / 1e6:6.0f} TF/s | bitwise equal {same}", flush=True)
    del a, w, out, ref
print(f"the ten store GEMMs of one forward: gemm_xs_kernel {tot_s:8.1f} us | gemm_xp_kernel {tot_p:8.1f} us", flush=True)
if "--forward" in sys.argv:
    import numpy as np
    from shapegen_amd import specs
    from shapegen_amd.diffusion import PointCloudDiffusion
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=1.3).items()}
    model = PointCloudDiffusion(num_points=2048)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    x = torch.randn(64, 2048, 3, device="cuda")
    t = torch.rand(64, device="cuda")
    outs = {}
    for rnd in range(4):
        for cfg, name in ((11, "gemm_xs_kernel"), (10, "gemm_xp_kernel")):
            lib.pcd_gemm_set_config(cfg)
            for _ in range(5): y = model.model(x, t)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): y = model.model(x, t)
            e1.record(); torch.cuda.synchronize()
            outs[name] = y.clone()
            print(f"forward, store GEMMs on {name}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    lib.pcd_gemm_set_config(11)
    print("forward outputs bitwise equal:", torch.equal(outs["gemm_xs_kernel"], outs["gemm_xp_kernel"]))
