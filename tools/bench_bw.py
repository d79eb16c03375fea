"""Dev tool: reference HBM write / copy bandwidth for buffers of the layer-output sizes."""
import torch
for mb in (67, 268, 537, 1074):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.float16, device="cuda"); b = torch.ones(n, dtype=torch.float16, device="cuda")
    for name, fn in (("fill", lambda: a.fill_(1.0)), ("copy", lambda: a.copy_(b))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"{mb} MB {name}: {t*1e3:.1f} us -> write {mb/t/1e3:.2f} TB/s", flush=True)
