"""Dev tool: time one training step (forward + backward + AdamW) of the point denoiser at the reference's training
configuration (batch 16 x 2048 points, train_point_ddpm.py:43-46) on one MI355X."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd.diffusion import PointCloudDiffusion
from shapegen_amd.training import PointTrainer

B, N = int(os.environ.get("B", 16)), int(os.environ.get("N", 2048))
torch.manual_seed(0)
model = PointCloudDiffusion(num_points=N).to("cuda")
tr = PointTrainer(model.model, lr=1e-4)
x0 = torch.randn(B, N, 3, device="cuda") * 0.4
steps = int(os.environ.get("STEPS", 10))
def one():
    t = torch.rand(B, device="cuda")
    x_t, noise, _, _ = model.add_noise(x0, t)
    return tr.train_step(x_t, t, noise)
for _ in range(3):
    loss = one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flop = 3 * 42_615_552 * B * N          # literal forward count x (forward + backward-data + backward-weight)
print(f"B={B} N={N}: {dt*1e3:.2f} ms/step  {1/dt:.1f} steps/s  {B/dt:.0f} shapes/s  ~{flop/dt/1e12:.0f} TFLOP/s dense-equivalent  loss {loss.item():.4f}")
