"""Dev tool: time one VAE3DLarge training step (forward + backward + Adam) at the reference's batch size 16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd.training_vae import VAETrainer
from shapegen_amd.vae import VAE3DLarge
B = int(os.environ.get("B", 16))
torch.manual_seed(0)
vae = VAE3DLarge().to("cuda")
tr = VAETrainer(vae, lr=1e-4)
x = (torch.rand(B, 1, 32, 32, 32, device="cuda") > 0.9).float()
for _ in range(2):
    out = tr.train_step(x, 0.01)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = int(os.environ.get("STEPS", 5))
for _ in range(steps):
    out = tr.train_step(x, 0.01)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flop = 3 * 2 * (12.24e9 + 20.10e9) * B
print(f"VAE3DLarge train step B={B}: {dt*1e3:.1f} ms  {B/dt:.0f} grids/s  ~{flop/dt/1e12:.0f} TFLOP/s dense-equivalent  loss {float(out[0]):.4f}  "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
