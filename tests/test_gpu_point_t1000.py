"""The point path at the horizon BASELINE configs[1] names: N = 2048 points, 1000 steps (reference defaults
diffusion.py:226,262,292; test_point_ddpm.py:36,78-92), against goldens captured from the reference itself
(G19-G21, `oracle/make_golden.py g19|g20|g20b|g21`).

Both arithmetic modes of the denoiser run every case:
  fp16 (product path: fp16 operands, fp32 accumulation; graph replay where the sampler uses it)
      cloud rel-L2 <= 2e-3 (the short-horizon tests' 5e-3, tightened here: measured 9.8e-5 / 3.1e-4 / 2.4e-4 with the hi / lo narrow
      weights), |CD_build - CD_ref| <= 1e-4 (scaling 1, north_star's gate)
  fp32 (SURVEY 8(c) parity mode, csrc/unet_f32.hip)
      cloud rel-L2 <= 5e-5 and max-abs <= 1e-3 for clouds up to |x| = 100 -- the survey's bound, stated for O(1-100) clouds --
      scaled with the cloud beyond that (max-abs <= 1e-5 max|x_ref|: 1e-3 absolute on a value of 1e3 would be below fp32's own
      resolution of the sums that produced it)
Intermediate states (the denoiser's input at calls 100, 250, ... 999 of the reference's loop) are compared too, so a
drift is located in time, not only seen at the end.  Measured values: profiles/r04_b_t1000_divergence.txt.

The DDPM sampler (`sample2`) has two fixtures.  With the synthetic weights every other fixture uses (gain 1.3) the
REFERENCE's own loop is unstable -- an untrained denoiser does not cancel the noise the ancestral update re-injects, and the
state reaches |x| = 9.4e8 by step 1000 -- so G20 is kept as the runaway record (the fp32 mode follows it to 1.2e-4
relative even there; the fp16 path is checked while the state is inside fp16's range).  G20b uses the same generator at
gain 1.0, where the state stays at the scale the loop's own noise accumulation gives (rms ~ 330): that one carries the bounds.
"""
import numpy as np
import pytest
import torch

from helpers import point_sd, rel_l2
from shapegen_amd import specs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

SINKHORN_REL = 3e-4        # Sinkhorn EMD of the reconstructed cloud against the reference's: 3 x the measured deviation (fp16 path 3.4e-5 ... 8.9e-5 over G21's four
                           # samples, fp32 mode 1.6e-6 ... 7.5e-6); was 2 % until round 5
TOL = {"fp16": dict(rel=2e-3, maxabs=None), "fp32": dict(rel=5e-5, maxabs=1e-3)}      # measured: fp16 9.8e-5 ... 3.1e-4, fp32 1.4e-6 ... 2.3e-6


@pytest.fixture(scope="module")
def models():
    from shapegen_amd.diffusion import PointCloudDiffusion
    out = {}
    for prec in ("fp16", "fp32"):
        m = PointCloudDiffusion(num_points=2048)
        m.load_state_dict(point_sd(), strict=True)
        m = m.to("cuda").eval()
        m.model.set_precision(prec)
        out[prec] = m
    return out


class HashedNoises:
    """noises[k] = the k-th per-step normal draw of the G20 capture, rebuilt from the integer hash (never stored)."""

    def __init__(self, tag, shape):
        self.tag, self.shape = tag, tuple(shape)

    def __getitem__(self, k):
        n = int(np.prod(self.shape))
        return torch.from_numpy(specs.hash_normal(f"{self.tag}{k}", n, 0).astype(np.float32).reshape(self.shape))


def spy_inputs(model, calls):
    """Record the state handed to the denoiser at the given call indices (eager stepping only)."""
    rec, n = {}, [0]
    inner = model.model.forward_with_bias

    def fwd(x, tb, stride, out=None):
        if n[0] in calls:
            rec[n[0]] = x.detach().clone().cpu()
        n[0] += 1
        return inner(x, tb, stride, out=out)

    return rec, fwd, inner


def check_cloud(prec, got, want, what):
    tol = TOL[prec]
    r = rel_l2(got, want)
    mx = float((torch.as_tensor(got) - torch.as_tensor(want)).abs().max())
    print(f"{what} [{prec}]: rel-L2 {r:.3e}  max-abs {mx:.3e}")
    assert r < tol["rel"], (what, prec, r)
    if tol["maxabs"] is not None:
        scale = max(1.0, float(torch.as_tensor(want).abs().max()) / 100.0)
        assert mx < tol["maxabs"] * scale, (what, prec, mx, scale)


def chamfer_gate(out, want, other, bound=1e-4):
    """north_star's quality gate: |CD_build - CD_ref| <= 1e-4 (scaling 1) against the same third cloud.  `chamfer_distance`
    normalises both clouds to the unit cube first (metrics.py:37-38), so the value does not depend on the clouds' scale."""
    from shapegen_amd import metrics as M
    cd_build = float(M.chamfer_distance(out, other, 1))
    cd_ref = float(M.chamfer_distance(want, other, 1))
    cd_pair = float(M.chamfer_distance(out, want, 1))
    print(f"   CD(gpu,ref)={cd_pair:.3e}  CD_build={cd_build:.6f}  CD_ref={cd_ref:.6f}  |dCD|={abs(cd_build - cd_ref):.2e}")
    assert abs(cd_build - cd_ref) < bound, (cd_build, cd_ref)
    return cd_pair


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_ddim_1000_steps_at_2048_points(models, golden, prec):
    """G19: `sample(2, 2048)` with the reference's default 1000 steps.  The product call (graph replay) gives the final
    cloud; a second, eager run with graphs off records the intermediate states and must end in the same cloud."""
    g = golden("point_t1000_ddim.npz")
    m = models[prec]
    xT = torch.from_numpy(g["xT"]).cuda()
    out = m.sample(2, 2048, x_T=xT)
    want = torch.from_numpy(g["out"])
    check_cloud(prec, out.cpu(), want, "DDIM T=1000 final x0")
    chamfer_gate(out, want.cuda(), xT)
    calls = [int(c) for c in g["ckpt_calls"]]
    rec, fwd, inner = spy_inputs(m, calls)
    m.model.forward_with_bias = fwd
    m.use_graphs = False
    try:
        out_eager = m.sample(2, 2048, x_T=xT)
    finally:
        m.model.forward_with_bias = inner
        del m.use_graphs
    assert rel_l2(out_eager.cpu(), out.cpu()) < 1e-6          # the replayed graphs enqueue what the eager loop enqueues
    for i, c in enumerate(calls):
        check_cloud(prec, rec[c], torch.from_numpy(g["ckpt_x"][i]), f"   state before call {c}")


def _models_with_gain(gain):
    from helpers import as_torch
    from shapegen_amd.diffusion import PointCloudDiffusion
    sd = as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=gain))
    out = {}
    for prec in ("fp16", "fp32"):
        m = PointCloudDiffusion(num_points=2048)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda").eval()
        m.model.set_precision(prec)
        out[prec] = m
    return out


def _run_ddpm(m, g):
    xT = torch.from_numpy(g["xT"]).cuda()
    calls = [int(c) for c in g["ckpt_calls"]]
    rec, fwd, inner = spy_inputs(m, calls)
    m.model.forward_with_bias = fwd
    try:
        out = m.sample2(2, 2048, x_T=xT, noises=HashedNoises("g20.z", (2, 2048, 3)))
    finally:
        m.model.forward_with_bias = inner
    return out, rec, calls, xT


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_ddpm_1000_steps_at_2048_points(golden, prec):
    """G20b: `sample2(2, 2048)`, the sampler bench.py times, 1000 steps, with the reference's 999 per-step normal draws
    rebuilt from the integer hash on both sides; synthetic weights at gain 1.0 (see the module docstring)."""
    g = golden("point_t1000_ddpm_stable.npz")
    assert int(g["n_draws"]) == 999 and float(g["gain"]) == 1.0
    m = _models_with_gain(1.0)[prec]
    out, rec, calls, xT = _run_ddpm(m, g)
    want = torch.from_numpy(g["out"])
    check_cloud(prec, out.cpu(), want, "DDPM T=1000 final x")
    # north_star's Chamfer gate on this trajectory too.  With plain fp16 weights everywhere the fp16 path measured |dCD| = 2.5e-4 here
    # (cloud rel-L2 1.45e-3): `tools/attribute_fp16_layers.py` traced 1.4e-3 of it to the fp16 rounding of the WEIGHTS of six narrow
    # layers (enc1.conv2/3, dec1.*, output.0), which now carry hi / lo weights (`UNetPointNetLarge.hilo_mask`).
    chamfer_gate(out, want.cuda(), xT)
    for i, c in enumerate(calls):
        check_cloud(prec, rec[c], torch.from_numpy(g["ckpt_x"][i]), f"   state before call {c}")


def test_ddpm_runaway_record(models, golden):
    """G20: the same loop with the gain-1.3 weights, where the reference itself runs away (|x| = 9.4e8 at the end).  The fp32
    parity mode follows the unstable trajectory (every per-step rounding difference is amplified with the state: <= 1e-3
    relative at the end, measured 1.2e-4); the fp16 path is held to its bound while the state is inside fp16's range (call 100:
    |x| max 374) and is NOT expected to follow beyond it: activations saturate at 65504 once |x| passes ~1e4."""
    g = golden("point_t1000_ddpm.npz")
    assert float(np.abs(g["out"]).max()) > 1e8
    out32, rec32, calls, _ = _run_ddpm(models["fp32"], g)
    r = rel_l2(out32.cpu(), g["out"])
    print(f"runaway DDPM, fp32 mode vs reference: final rel-L2 {r:.3e}")
    assert r < 1e-3
    for i, c in enumerate(calls):
        assert rel_l2(rec32[c], g["ckpt_x"][i]) < 1e-3, c
    _, rec16, _, _ = _run_ddpm(models["fp16"], g)
    i100 = calls.index(100)
    r16 = rel_l2(rec16[100], g["ckpt_x"][i100])
    print(f"runaway DDPM, fp16 path vs reference at call 100 (|x| max {float(np.abs(g['ckpt_x'][i100]).max()):.0f}): rel-L2 {r16:.3e}")
    assert r16 < 5e-3


def synth_cloud(b, n, seed):
    """The clouds `oracle/make_golden.py` fed the reference (grid-like, unit-sphere normalised; data.py:213-254 shaped)."""
    out = np.zeros((b, n, 3), np.float32)
    for i in range(b):
        u = specs.hash_uniform(f"cloud{i}", 3 * 4096, seed).reshape(-1, 3)
        blob = np.round((u * 0.5 + 0.5) * np.array([31, 15, 9]) + np.array([0, 8, 11]))
        pts = np.unique(blob, axis=0)
        pts = pts - pts.mean(0)
        pts = pts / np.max(np.linalg.norm(pts, axis=1))
        sel = (specs.hash_uniform(f"sel{i}", n, seed) * 0.5 + 0.5) * len(pts)
        out[i] = pts[np.clip(sel.astype(np.int64), 0, len(pts) - 1)]
    return out


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_reconstruction_flow_with_metrics_at_2048_points(models, golden, prec):
    """G21: test_point_ddpm.py:74-92 at (4, 2048): add_noise(t = 0.01) -> sample3 (1000 steps) -> compute_metrics per
    sample.  add_noise is bit exact; the reconstructed clouds and the (Chamfer x1e3, Hungarian EMD, voxel BCE) triples
    are held to the reference's.  Metric bounds: Chamfer |d| <= 0.1 (= 1e-4 at scaling 1); EMD 1 % (an assignment cost,
    continuous in the points); BCE: one voxel of 32768 flipping costs 100/32768 = 3.05e-3, and a coordinate that sits
    within the cloud tolerance of a cell boundary may flip -- at most 8 cells per cloud are allowed to."""
    from shapegen_amd import metrics as M
    g = golden("point_t1000_recon.npz")
    m = models[prec]
    B = 4
    x0 = torch.from_numpy(synth_cloud(B, 2048, int(g["seed_cloud"]))).cuda()
    t = torch.ones(B, device="cuda") * 0.010
    eps = torch.from_numpy(specs.hash_normal("g21.eps", B * 2048 * 3, 0).astype(np.float32).reshape(B, 2048, 3)).cuda()
    noisy, _, nr, sr = m.add_noise(x0, t, noise=eps)
    assert np.array_equal(np.array([float(nr[0]), float(sr[0])], np.float32), g["add_rates"])
    assert torch.equal(noisy.cpu(), torch.from_numpy(g["noisy"]))
    out = m.sample3(num_samples=B, num_points=2048, x=noisy, start_t=t)
    want = torch.from_numpy(g["out"])
    check_cloud(prec, out.cpu(), want, "sample3 T=1000 reconstruction")
    for i in range(B):
        cd, emd, rec = (float(v) for v in M.compute_metrics(x0[i], out[i]))
        rcd, remd, rrec = g["triples"][i]
        print(f"   sample {i}: cd {cd:.4f} / {rcd:.4f}   emd {emd:.5f} / {remd:.5f}   recon {rec:.4f} / {rrec:.4f}")
        assert abs(cd - rcd) < 0.1
        assert abs(emd - remd) < 1e-2 * remd
        assert abs(rec - rrec) <= 8 * 100.0 / 32768 + 1e-6
        _, semd, _ = (float(v) for v in M.compute_metrics(x0[i], out[i], use_approximate_gpu_emd=True))
        sref = float(g["triples_sinkhorn"][i][1])
        print(f"      sinkhorn emd {semd:.6f} / {sref:.6f}: relative deviation {abs(semd - sref) / abs(sref):.2e}")
        assert abs(semd - sref) < SINKHORN_REL * abs(sref) + 1e-6
        cd1 = float(M.chamfer_distance(x0[i], out[i], 1))
        assert abs(cd1 - g["cd_s1"][i]) < 1e-4
