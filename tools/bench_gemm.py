"""Dev tool: time the GEMM tile configurations on the point-UNet layer shapes (MI355X only).
Interleaved rounds in one process (cdna guide rule 24); post-ReLU-like fp16 data (half zeros)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib, ops

lib = _lib.load()
M = 64 * 2048
shapes = [(2048, 4096, "colmax"), (1024, 2048, "f16"), (1024, 1024, "f16"), (512, 1024, "f16"), (512, 512, "f16"),
          (256, 512, "f16"), (256, 256, "f16"), (128, 128, "f16"), (64, 128, "f16")]
cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,4".split(","))]
g = torch.Generator(device="cuda").manual_seed(0)
for K, C, kind in shapes:
    a = torch.randn(M, K, device="cuda", generator=g).clamp_min(0).half()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    out = torch.empty(M, C, dtype=torch.float16, device="cuda") if kind == "f16" else None
    ref = None
    res = {}
    for rnd in range(3):
        for cfg in cfgs:
            if cfg == 0 and C > 64:
                continue
            lib.pcd_gemm_set_config(cfg)
            fn = (lambda: ops.gemm_f16(a, w, bias, relu=True, out=out)) if kind == "f16" else (lambda: ops.gemm_f16_colmax(a, w, bias, 2048))
            r = fn()
            if rnd == 0:
                rr = r.float().clone()
                if ref is None:
                    ref = rr
                else:
                    err = float((rr - ref).abs().max())
                    assert err < 1e-2 * float(ref.abs().max()) + 1e-3, (cfg, err)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(cfg, []).append(e0.elapsed_time(e1) / 5)
    line = f"K={K:5d} C={C:5d} {kind:6s}"
    for cfg in cfgs:
        if cfg in res:
            t = min(res[cfg])
            line += f" | cfg{cfg}: {t*1e3:8.1f}us {2.0*M*K*C/t/1e9:7.0f}TF"
    print(line, flush=True)
lib.pcd_gemm_set_config(-1)
