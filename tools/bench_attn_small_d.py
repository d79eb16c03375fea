"""Dev tool: the set-attention kernel at d = 32 / 16 (B = 64, N = 2048, 4 heads): the one-block kernel (set_attention_om_kernel), the two-block software pipeline
(set_attention_spn_kernel) and the pipeline's timing ablations (outputs wrong while set), min of 3 x 10 launches after a 20-launch ramp, one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import _lib
lib = _lib.load()
B, N, H = 64, 2048, 4
LEGS = [("one block per wave (om)", [3]), ("two blocks per wave (spn)", [4]), ("  spn, no K/V restaging", [4, 17]), ("  spn, restaging from four resident tiles", [4, 24]), ("  spn, no rare-path test", [4, 18]),
        ("  spn, neither", [4, 19]), ("  spn, neither, no waits / barriers", [4, 23])]
for C in (256, 128, 64):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B * N, 3 * C, device="cuda", generator=g) * 0.7).half()
    out = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
    def fn():
        _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B, N, C, H, out.data_ptr(), 0, 0, _lib.stream_ptr()))
    for name, cfgs in (LEGS if C < 256 else [("two blocks per wave (sp)", [4, 5]), ("  sp, eight waves per workgroup", [4, 6]), ("two blocks per wave (sp)", [4, 5]), ("  sp, eight waves per workgroup", [4, 6]), ("  sp, no K/V restaging", [4, 17]), ("  sp, restaging from four resident tiles", [4, 24]), ("  sp, no rare-path test", [4, 18]), ("  sp, neither", [4, 19])]):
        _lib.check(lib.pcd_set_attention_config(16))
        for c in cfgs:
            _lib.check(lib.pcd_set_attention_config(c))
        for _ in range(20):
            fn()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        fl = 4.0 * B * N * N * C
        print(f"d={C // H:2d} {name:40s} {best * 1e3:7.1f} us  {fl / best / 1e9:5.0f} TFLOP/s ({fl / best / 1e9 / 25:.1f} % of 2.5 PF)", flush=True)
    _lib.check(lib.pcd_set_attention_config(16)); _lib.check(lib.pcd_set_attention_config(4)); _lib.check(lib.pcd_set_attention_config(5))
