"""Dev tool: SURVEY 8(d) cfg4 end to end on one MI355X: encode 32 voxel grids -> 1000-step latent DDIM `sample` from
the encoded latents' shape -> VAE decode -> point clouds.  Synthetic weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import shapegen_amd
from shapegen_amd import specs
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge

B, T = 32, int(os.environ.get("T", 1000))
torch.manual_seed(24)
sd = specs.synth_state_dict(specs.latent_unet_spec(prefix="model."), seed=0, gain=1.3)
sd.update(specs.synth_state_dict(specs.vae3d_large_spec(prefix="vae."), seed=0, gain=1.3))
vae = VAE3DLarge()
m = LatentDiffusion(vae)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.to("cuda").eval()
m.use_graphs = bool(int(os.environ.get("USE_GRAPHS", "1")))
if "GRAPH_STEPS" in os.environ:
    m.GRAPH_STEPS = int(os.environ["GRAPH_STEPS"])
vox = (torch.rand(B, 1, 32, 32, 32, device="cuda") > 0.9).float()
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu, logvar = m.vae.encode(vox)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    clouds = m.sample(num_samples=B, num_steps=T)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, clouds
run()
enc, samp, clouds = run()
print(f"cfg4 B={B} T={T}: encode {enc*1e3:.2f} ms, sample (T latent steps + decode + voxel->points) {samp*1e3:.1f} ms "
      f"= {T/samp:.0f} denoising-steps/s; clouds {[int(c.shape[0]) for c in clouds[:4]]}...")
