"""Dev tool: timeline of the persistent latent kernel (instrumented build, s_memrealtime stamps at 100 MHz): for every phase of
one steady-state step, when its units started waiting, when their operands were in, when their stores had been acknowledged.
Prints, per phase: first / last "operands in" and "stored" relative to the step's start, and the gap to the previous phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import shapegen_amd
from shapegen_amd import _lib
from shapegen_amd.diffusion import LatentDiffusion
from shapegen_amd.vae import VAE3DLarge
from helpers import latent_sd
torch.set_grad_enabled(False)
lib = _lib.load()
m = LatentDiffusion(VAE3DLarge()); m.load_state_dict(latent_sd(), strict=True); m = m.to("cuda").eval()
B, T, STEPS = 32, 64, 48
h, _ = m.model._persist_handle()
buf = torch.zeros(256 * STEPS * 8 * 8, dtype=torch.int32, device="cuda")
lib.pcd_latent_persist_trace(h, buf.data_ptr(), STEPS)
zT = torch.randn(B, 256, device="cuda")
m.sample(B, num_steps=T, z_T=zT)
m.sample(B, num_steps=T, z_T=zT)
torch.cuda.synchronize()
lib.pcd_latent_persist_trace(h, 0, 0)
tr = buf.cpu().numpy().astype(np.int64).reshape(256, STEPS, 8, 8) & 0xffffffff
import ctypes
plan = (ctypes.c_int * (256 * 8))()
lib.pcd_latent_persist_plan_dump(plan)
plan = np.array(plan).reshape(256, 8)
names = ["enc1", "enc2", "enc3", "enc4", "gf0", "gf3", "dec4", "dec3", "dec2", "dec1", "out0", "out2"]
for step in (40, 41):
    ent, inn, pre, out = tr[:, step, :, 0], tr[:, step, :, 1], tr[:, step, :, 3], tr[:, step, :, 2]
    first = int(plan[plan >= 0].min())                     # the step's first phase (the enc1 -> enc2 -> enc3 chain)
    t0 = ent[plan == first].min()
    print(f"step {step}: times in us from the first unit's entry; per phase: units | waiting from (median) | operands in first..last | "
          f"epilogue computed (median after operands) | stores acknowledged first..last")
    prev = 0.0
    for ph in sorted(set(plan[plan >= 0].tolist())):
        sel = plan == ph
        e, i, p_, o = (ent[sel] - t0) / 100.0, (inn[sel] - t0) / 100.0, (pre[sel] - t0) / 100.0, (out[sel] - t0) / 100.0
        print(f"  {names[ph // 2]:5s}{' fin' if ph & 1 else '    '} {sel.sum():4d} | {np.median(e):6.2f} | {i.min():6.2f}..{i.max():6.2f} | +{np.median(p_ - i):4.2f} | "
              f"{o.min():6.2f}..{o.max():6.2f}   (+{i.max() - prev:5.2f} after the previous phase's last ack)")
        prev = o.max()
        wv = (tr[:, step, :, 4:8][sel] - t0) / 100.0            # operands in, per wave (chains: stage stamps in 5 .. 7)
        if ph in (4, 22) and first != 0:                       # (an experimental build fused enc1-enc3 / dec1-output.2 into chain units)
            print(f"        chain stages (median): operands in {np.median(wv[:, 0]):6.2f} | first layer done {np.median(wv[:, 1]):6.2f} | second {np.median(wv[:, 2]):6.2f} | "
                  f"last layer's products {np.median(wv[:, 3]):6.2f} | epilogue computed {np.median(p_):6.2f}")
            continue
        print(f"        per wave operands in (median over units): " + " ".join(f"{np.median(wv[:, k]):6.2f}" for k in range(4)) +
              f" | slowest wave - fastest wave, median {np.median(wv.max(1) - wv.min(1)):.2f} max {(wv.max(1) - wv.min(1)).max():.2f}")
        if i.max() - i.min() > 2.0 and step == 41:
            ws = np.argwhere(sel)
            late = sorted(((inn[w, u] - t0) / 100.0, (ent[w, u] - t0) / 100.0, int(w)) for w, u in ws)
            print("      late operands (in, entered, vwg):", " ".join(f"{a:.1f}/{b:.1f}/{w}" for a, b, w in late[::max(1, len(late) // 24)]))
    t_next = tr[:, step + 1, :, 0][plan == first].min()
    print(f"  step length {(t_next - t0) / 100.0:.2f} us")
