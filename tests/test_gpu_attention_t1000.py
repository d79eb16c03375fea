"""The attention denoiser under the reference's sampling loops at the horizon BASELINE configs[1] names (G26, `oracle/make_golden.py g26a | g26b`): the reference's
`PointCloudDiffusion.sample` / `sample2` (diffusion.py:225-289) with `pcd.model` replaced by `UNetAttentionPointExperimental(2048)` (networks.py:597-722), (2, 2048),
1000 steps; start noise and the 999 DDPM draws from the integer hash, the state handed to the denoiser at calls 100 ... 999 stored like G19 / G20.  Until round 5 this
backbone -- the carrier of north_star's set-attention target -- was pinned to the reference per forward only (G10, G24) and under the samplers only against the oracle
at N = 128, T <= 100.

Both arithmetic modes: fp16 (product path) and fp32 (parity mode, csrc/attn_f32.hip).  Bounds next to the measured values in `TOL`; north_star's Chamfer gate
|CD_build - CD_ref| <= 1e-4 on every trajectory.  The DDPM fixture uses the synthetic weights at gain 0.6: at the 1.0 every other attention fixture uses the reference's
own loop runs away over this untrained denoiser (rms 3.8e3 at call 100, NaN before call 900 -- SURVEY A.9; make_golden.py records why)."""
import numpy as np
import pytest
import torch

from helpers import as_torch, rel_l2
from shapegen_amd import specs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# the point path's bounds.  Measured (round 5): fp16 cloud rel-L2 4.5e-4 (DDIM) / 4.3e-4 (DDPM), |dCD| 2.8e-6 / 2.4e-5 against the 1e-4 gate, growing smoothly over the
# horizon (3.8e-4 at call 100 -> 4.4e-4 at call 999; DDPM 1.9e-5 -> 4.3e-4); fp32 mode 2.2e-6 / 7.9e-7.  The per-forward 1.5e-3 of G24 does not compound: no hi / lo
# weights were needed on this backbone.
TOL = {"fp16": dict(rel=2e-3), "fp32": dict(rel=5e-5)}


def hashed(tag, shape):
    return torch.from_numpy(specs.hash_normal(tag, int(np.prod(shape)), 0).astype(np.float32).reshape(tuple(shape)))


class HashedNoises:
    def __init__(self, tag, shape):
        self.tag, self.shape = tag, tuple(shape)

    def __getitem__(self, k):
        return hashed(f"{self.tag}{k}", self.shape)


def build(gain, prec):
    from shapegen_amd.diffusion import PointCloudDiffusion
    m = PointCloudDiffusion(num_points=2048, backbone="attention")
    # (the synthetic generator hashes the KEY: generate under the names the reference capture used -- `UNetAttentionPointExperimental`'s own -- then prefix)
    sd = as_torch(specs.synth_state_dict(specs.unet_attention_spec(), seed=0, gain=gain))
    m.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    m.model.set_precision(prec)
    return m


def spy_inputs(model, calls):
    rec, n = {}, [0]
    inner = model.model.forward_with_bias

    def fwd(x, tb, stride, out=None):
        if n[0] in calls:
            rec[n[0]] = x.detach().clone().cpu()
        n[0] += 1
        return inner(x, tb, stride, out=out)

    return rec, fwd, inner


def check_cloud(prec, got, want, what):
    r = rel_l2(got, want)
    mx = float((torch.as_tensor(got) - torch.as_tensor(want)).abs().max())
    print(f"{what} [{prec}]: rel-L2 {r:.3e}  max-abs {mx:.3e}  (max |ref| {float(torch.as_tensor(want).abs().max()):.3g})")
    assert r < TOL[prec]["rel"], (what, prec, r)
    return r


def chamfer_gate(out, want, other, bound=1e-4):
    from shapegen_amd import metrics as M
    cd_build = float(M.chamfer_distance(out, other, 1))
    cd_ref = float(M.chamfer_distance(want, other, 1))
    print(f"   CD(gpu,ref)={float(M.chamfer_distance(out, want, 1)):.3e}  CD_build={cd_build:.6f}  CD_ref={cd_ref:.6f}  |dCD|={abs(cd_build - cd_ref):.2e}")
    assert abs(cd_build - cd_ref) < bound, (cd_build, cd_ref)


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_attention_backbone_ddim_1000_steps_at_2048_points(golden, prec):
    """G26a: `sample(2, 2048)` over UNetAttentionPointExperimental, the reference's default 1000 steps.  The product call (graph replay) gives the final cloud; an eager
    run records the states handed to the denoiser at the reference's checkpoint calls and must end in the same cloud."""
    g = golden("attention_t1000_ddim.npz")
    assert float(g["gain"]) == 1.0 and int(g["n_draws"]) == 0
    m = build(1.0, prec)
    xT = hashed("g26a.xT", (2, 2048, 3))
    assert np.array_equal(g["ckpt_x"][0], xT.numpy())               # the start noise the reference drew IS the hashed tensor
    out = m.sample(2, 2048, x_T=xT.cuda())
    want = torch.from_numpy(g["out"])
    check_cloud(prec, out.cpu(), want, "attention backbone, DDIM T=1000 final x0")
    chamfer_gate(out, want.cuda(), xT.cuda())
    calls = [int(c) for c in g["ckpt_calls"]]
    rec, fwd, inner = spy_inputs(m, calls)
    m.model.forward_with_bias = fwd
    m.use_graphs = False
    try:
        out_eager = m.sample(2, 2048, x_T=xT.cuda())
    finally:
        m.model.forward_with_bias = inner
        del m.use_graphs
    assert rel_l2(out_eager.cpu(), out.cpu()) < 1e-6
    for i, c in enumerate(calls):
        check_cloud(prec, rec[c], torch.from_numpy(g["ckpt_x"][i]), f"   state before call {c}")


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_attention_backbone_ddpm_1000_steps_at_2048_points(golden, prec):
    """G26b: `sample2(2, 2048)` over UNetAttentionPointExperimental, 1000 steps, the reference's 999 draws of `torch.randn_like` rebuilt from the integer hash."""
    g = golden("attention_t1000_ddpm.npz")
    assert int(g["n_draws"]) == 999 and float(g["gain"]) == 0.6
    want = torch.from_numpy(g["out"])
    assert torch.isfinite(want).all() and float(want.abs().max()) < 5e3
    m = build(0.6, prec)
    xT = hashed("g26b.xT", (2, 2048, 3))
    assert np.array_equal(g["ckpt_x"][0], xT.numpy())
    calls = [int(c) for c in g["ckpt_calls"]]
    rec, fwd, inner = spy_inputs(m, calls)
    m.model.forward_with_bias = fwd
    try:
        out = m.sample2(2, 2048, x_T=xT.cuda(), noises=HashedNoises("g26b.z", (2, 2048, 3)))
    finally:
        m.model.forward_with_bias = inner
    check_cloud(prec, out.cpu(), want, "attention backbone, DDPM T=1000 final x")
    chamfer_gate(out, want.cuda(), xT.cuda())
    for i, c in enumerate(calls):
        check_cloud(prec, rec[c], torch.from_numpy(g["ckpt_x"][i]), f"   state before call {c}")
