"""Dev tool: time the latent denoiser step (SimpleLatentUNetPointNet forward + DDIM update) and VAE decode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import shapegen_amd
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
for B in (32, 256):
    z = torch.randn(B, 256, device="cuda")
    from shapegen_amd.diffusion import Stepper
    tab = m.ddim_table(1000, B)
    stp = Stepper(m, z, tab, m.model.time_bias(tab.t), m._forward_fn(), "ddim")
    for k in range(5):
        stp.step(k, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 200
    for k in range(K):
        stp.step(k, True)
    torch.cuda.synchronize(); dt_e = (time.perf_counter() - t0) / K
    stp.capture()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        stp.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"  eager {dt_e*1e6:.1f} us/step, graph {dt*1e6:.1f} us/step")
    print(f"latent step B={B}: {dt*1e6:.1f} us/step  -> {1/dt:.0f} steps/s ; weight-stream roofline 38.2MB/8TB/s = 4.8us", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    zz = torch.randn(min(B, 32), 256, device="cuda")
    m.vae.decode(zz); torch.cuda.synchronize()
    e0.record(); out = m.vae.decode(zz); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"  vae.decode B={zz.shape[0]}: {ms:.2f} ms -> {40.2e9*zz.shape[0]/ms/1e9:.0f} TFLOP/s", flush=True)
    vox = (torch.rand(zz.shape[0], 1, 32, 32, 32, device="cuda") > 0.9).float()
    m.vae.encode(vox); torch.cuda.synchronize()
    e0.record(); mu, lv = m.vae.encode(vox); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"  vae.encode B={zz.shape[0]}: {ms:.2f} ms -> {24.48e9*zz.shape[0]/ms/1e9:.0f} TFLOP/s", flush=True)
