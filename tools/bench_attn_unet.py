"""Dev tool: time one forward of UNetAttentionPointExperimental (a10) at B=64, N=2048 on one MI355X."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import shapegen_amd
from shapegen_amd import specs
from shapegen_amd.networks import UNetAttentionPointExperimental
torch.set_grad_enabled(False)
B, N = int(os.environ.get("B", 64)), 2048
net = UNetAttentionPointExperimental(num_points=N)
sd = specs.synth_state_dict(specs.unet_attention_spec(), seed=0, gain=1.0)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
net = net.to("cuda").eval()
x = torch.randn(B, N, 3, device="cuda") * 0.5
t = torch.full((B,), 0.5, device="cuda")
for _ in range(3):
    y = net(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    y = net(x, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
attn_flop = 4.0 * B * N * N * (64 + 128 + 256 + 256 + 256 + 128 + 64)
print(f"UNetAttentionPointExperimental forward B={B} N={N}: {dt*1e3:.2f} ms ({1/dt:.0f} forwards/s); attention products {attn_flop/1e12:.2f} TFLOP -> >= {attn_flop/dt/1e12:.0f} TFLOP/s overall")
