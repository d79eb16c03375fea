"""Pins the CPU oracle (oracle/torch_oracle.py) against golden vectors captured from the
imported reference (oracle/make_golden.py).  Tolerance: same ATen ops => <=1e-6 relative;
schedule tables bit-exact."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from shapegen_amd import specs
from helpers import point_sd, latent_sd, sab_sd, una_sd, rel_l2, voxels_from_idx, synth_voxels

torch.set_grad_enabled(False)


def test_schedule_tables_bit_exact(golden):
    g = golden("schedule.npz")
    n, s = O.offset_cosine_schedule(torch.from_numpy(g["cos_t"]))
    assert np.array_equal(n.numpy(), g["cos_noise"]) and np.array_equal(s.numpy(), g["cos_signal"])
    # SURVEY A.1 probe values
    assert abs(g["cos_noise"][-1] - 0.99979997) < 1e-7 and abs(g["cos_signal"][2] - 0.59447980) < 1e-7
    n, s = O.linear_schedule(torch.from_numpy(g["lin_t"]))
    assert np.array_equal(n.numpy(), g["lin_noise"]) and np.array_equal(s.numpy(), g["lin_signal"])


@pytest.mark.parametrize("T", [50, 100, 1000])
def test_sampler_time_sequences_bit_exact(golden, T):
    g = golden("schedule.npz")
    dummy = lambda x, t: torch.zeros_like(x)
    tr = []
    O.ddim_sample(dummy, torch.zeros(1, 1, 3), T, trace=tr)
    assert np.array_equal(np.asarray(tr, np.float32), g[f"sample_T{T}"])
    tr = []
    O.ddpm_sample(dummy, torch.zeros(1, 1, 3), T, [torch.zeros(1, 1, 3)] * (T - 1), trace=tr)
    got, want = np.asarray(tr, np.float32), g[f"sample2_T{T}"].copy()
    # column 3 of the golden holds sqrt(n_prev/n); the oracle trace holds n_prev
    want_coef = want[:-1, 3]
    got_coef = np.sqrt(got[:-1, 3] / got[:-1, 1]).astype(np.float32)
    np.testing.assert_allclose(got_coef, want_coef, rtol=2e-7)
    assert np.array_equal(got[:, [0, 1, 2]], want[:, [0, 1, 2]])
    assert np.array_equal(got[:-1, 4], want[:-1, 4])


def test_sample3_time_sequence_bit_exact(golden):
    g = golden("schedule.npz")
    dummy = lambda x, t: torch.zeros_like(x)
    for key, T, start in (("sample3_T1000_from0.01", 1000, 0.01), ("sample3_T100_from1", 100, 1.0)):
        tr = []
        O.ddim_from_state(dummy, torch.zeros(2, 1, 3), torch.ones(2) * start, T, trace=tr)
        assert np.array_equal(np.asarray(tr, np.float32), g[key], equal_nan=True)


def test_time_embedding(golden):
    g = golden("point_unet.npz")
    sd = point_sd()
    emb = O.timestep_embedding(torch.from_numpy(g["temb_t"]), 256)
    assert np.array_equal(emb.numpy(), g["temb_sin"])
    np.testing.assert_allclose(O.time_mlp(sd, "model.", emb).numpy(), g["temb_mlp"], rtol=1e-6, atol=1e-6)


def test_point_unet_forward_and_taps(golden):
    g = golden("point_unet.npz")
    sd = point_sd()
    taps = {}
    eps = O.unet_pointnet_large(sd, "model.", torch.from_numpy(g["fw_small_x"]),
                                torch.from_numpy(g["fw_small_t"]), taps=taps)
    assert rel_l2(eps, g["fw_small_eps"]) < 1e-6
    for mine, ref in (("x1", "enc1"), ("x2", "enc2"), ("x3", "enc3"), ("x4", "enc4"), ("pooled", "pooled"),
                      ("d4", "dec4"), ("d3", "dec3"), ("d2", "dec2"), ("d1", "dec1")):
        assert rel_l2(taps[mine], g["fw_small_" + ref]) < 1e-6, mine
    eps = O.unet_pointnet_large(sd, "model.", torch.from_numpy(g["fw_mid_x"]), torch.from_numpy(g["fw_mid_t"]))
    assert rel_l2(eps, g["fw_mid_eps"]) < 1e-6
    assert float(torch.from_numpy(g["fw_mid_eps"]).std()) > 0.05  # the fixture is not degenerate


def test_single_steps(golden):
    g = golden("point_unet.npz")
    x, eps, z = map(torch.from_numpy, (g["fw_small_x"], g["fw_small_eps"], g["step_z"]))
    n, s, nn_, sn = [torch.full((2,), float(v)) for v in g["step_rates"]]
    x0 = O.remove_noise(x, eps, n, s)
    np.testing.assert_allclose(x0.numpy(), g["step_x0"], rtol=1e-6, atol=1e-6)
    ddim = sn.view(-1, 1, 1) * x0 + nn_.view(-1, 1, 1) * eps
    np.testing.assert_allclose(ddim.numpy(), g["step_ddim"], rtol=1e-6, atol=1e-6)
    ddpm = sn.view(-1, 1, 1) * x0 + torch.sqrt(nn_ / n).view(-1, 1, 1) * n.view(-1, 1, 1) * z
    np.testing.assert_allclose(ddpm.numpy(), g["step_ddpm"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("T", [5, 50])
def test_ddim_sampler(golden, T):
    g = golden("point_samplers.npz")
    sd = point_sd()
    model = lambda x, t: O.unet_pointnet_large(sd, "model.", x, t)
    out = O.ddim_sample(model, torch.from_numpy(g[f"sample_T{T}_xT"]), T)
    assert rel_l2(out, g[f"sample_T{T}_out"]) < 1e-5


def test_ddpm_and_from_state_samplers(golden):
    g = golden("point_samplers.npz")
    sd = point_sd()
    model = lambda x, t: O.unet_pointnet_large(sd, "model.", x, t)
    out = O.ddpm_sample(model, torch.from_numpy(g["s2_xT"]), 20, list(torch.from_numpy(g["s2_z"])))
    assert rel_l2(out, g["s2_out"]) < 1e-5
    noisy, n, s = O.add_noise(torch.from_numpy(g["s3_x0"]), torch.ones(2) * 0.01, torch.from_numpy(g["s3_noise"]))
    np.testing.assert_allclose(noisy.numpy(), g["s3_noisy"], rtol=1e-6, atol=1e-7)
    out = O.ddim_from_state(model, noisy, torch.ones(2), 20)
    assert rel_l2(out, g["s3_T20_from1_out"]) < 1e-5


def test_latent_unet_and_vae(golden):
    g = golden("latent.npz")
    sd = latent_sd()
    eps = O.latent_unet(sd, "model.", torch.from_numpy(g["lat_z"]), torch.from_numpy(g["lat_t"]))
    assert rel_l2(eps, g["lat_eps"]) < 1e-6
    vox = voxels_from_idx([g["vae_occ_idx"], g["vae_occ_idx1"]])
    mu, logvar = O.vae_encode(sd, "vae.", vox, specs.VAE_ENC)
    assert rel_l2(mu, g["vae_mu"]) < 1e-5 and rel_l2(logvar, g["vae_logvar"]) < 1e-5
    dec = O.vae_decode(sd, "vae.", torch.from_numpy(g["vae_mu"]), specs.VAE_DEC)
    assert rel_l2(dec, g["vae_dec"]) < 1e-5
    for thr in (0.4, 0.5):
        pcs = O.voxel_tensor_to_point_clouds(torch.from_numpy(g["vae_dec"]), thr)
        for i, pc in enumerate(pcs):
            assert np.array_equal(pc.numpy(), g[f"v2p_thr{thr}_{i}"])  # integer-derived: bit exact


def test_latent_ddim(golden):
    g = golden("latent.npz")
    sd = latent_sd()
    model = lambda z, t: O.latent_unet(sd, "model.", z, t)
    z0 = O.ddim_sample(model, torch.from_numpy(g["ldm_T5_zT"]), 5)
    assert rel_l2(z0, g["ldm_T5_z0"]) < 1e-5
    pcs = O.voxel_tensor_to_point_clouds(O.vae_decode(sd, "vae.", z0, specs.VAE_DEC), 0.4)
    assert [len(p) for p in pcs] == list(g["ldm_T5_counts"])
    assert np.array_equal(pcs[0].numpy(), g["ldm_T5_pc0"])


def test_ddim_sampler_at_baseline_point_count(golden):
    """G18 (`make_golden.py n2048`): `sample(2, 2048, num_steps=50)` of the reference at BASELINE's N = 2048."""
    g = golden("point_n2048.npz")
    sd = point_sd()
    out = O.ddim_sample(lambda x, t: O.unet_pointnet_large(sd, "model.", x, t), torch.from_numpy(g["xT"]), 50)
    assert rel_l2(out, g["out"]) < 1e-5


def test_ddim_sampler_rows_of_the_full_launch_batch(golden):
    """G25 (`make_golden.py g25`): `sample(64, 2048, num_steps=50)` of the reference, start noise from the integer hash.  Shapes are independent in eval
    mode, so the oracle re-runs three of the 64 on their own and must land on the reference's rows of the batched run."""
    g = golden("point_b64.npz")
    xT = torch.from_numpy(specs.hash_normal("g25.xT", 64 * 2048 * 3, 0).astype(np.float32).reshape(64, 2048, 3))
    sd = point_sd()
    rows = [0, 37, 63]
    out = O.ddim_sample(lambda x, t: O.unet_pointnet_large(sd, "model.", x, t), xT[rows], 50)
    for k, r in enumerate(rows):
        assert rel_l2(out[k], g["out"][r]) < 1e-5, r


def test_cfg4_launch_shape_rows(golden):
    """G17 (`make_golden.py cfg4`): the reference at BASELINE configs[3]'s shape, B = 32, T = 1000.  Every sample is
    independent (GroupNorm per sample, eval BatchNorm3d), so the oracle runs the four decoded rows only."""
    g = golden("cfg4.npz")
    sd = latent_sd()
    rows = g["dec_rows"]
    vox = synth_voxels(32, 4)
    assert np.array_equal(vox.reshape(32, -1).sum(1).numpy().astype(np.int64), g["vox_counts"])
    mu, logvar = O.vae_encode(sd, "vae.", vox[rows], specs.VAE_ENC)
    assert rel_l2(mu, g["enc_mu"][rows]) < 1e-5 and rel_l2(logvar, g["enc_logvar"][rows]) < 1e-5
    z0 = O.ddim_sample(lambda z, t: O.latent_unet(sd, "model.", z, t), torch.from_numpy(g["zT"][rows]), 1000)
    assert rel_l2(z0, g["z0"][rows]) < 1e-4                      # 1000 dependent fp32 steps, row-batched differently
    dec = O.vae_decode(sd, "vae.", torch.from_numpy(g["z0"][rows]), specs.VAE_DEC)
    assert float((dec - torch.from_numpy(g["dec"]).float()).abs().max()) < 1e-3          # golden stored as fp16
    assert [int((d > 0.4).sum()) for d in dec] == list(g["counts"][rows])
    dm = O.vae_decode(sd, "vae.", torch.from_numpy(g["enc_mu"][rows]), specs.VAE_DEC)
    assert float((dm - torch.from_numpy(g["dec_of_mu"]).float()).abs().max()) < 1e-3


def test_metrics_units_known_answers(golden):
    """reference units.py:8-26 inputs; SURVEY section 4 probe values."""
    g = golden("metrics.npz")
    x, y = torch.from_numpy(g["units_x"]), torch.from_numpy(g["units_y"])
    assert abs(float(g["units_cd"]) - 142.7139) < 1e-3
    assert abs(float(g["units_emd_cpu"]) - 44.3071) < 1e-3
    assert abs(float(g["units_emd_sinkhorn"]) - 5.9952) < 1e-3
    assert abs(float(O.chamfer_distance(x, y)) - float(g["units_cd"])) < 1e-3
    assert abs(float(O.earth_mover_distance_cpu(x, y)) - float(g["units_emd_cpu"])) < 1e-4
    assert abs(float(O.earth_mover_distance_sinkhorn(x, y)) - float(g["units_emd_sinkhorn"])) < 1e-4
    assert abs(float(O.chamfer_distance_exact(x, y)) - float(g["units_cd"])) < 1e-2


def test_metrics_fixtures(golden):
    g = golden("metrics.npz")
    a, b = torch.from_numpy(g["m_a"]), torch.from_numpy(g["m_b"])
    assert np.array_equal(O.normalize_to_cube(a).numpy(), g["m_norm_a"])
    assert abs(float(O.chamfer_distance(a, b)) - float(g["m_cd_batch"])) < 1e-3
    assert abs(float(O.chamfer_distance(a, b, 1)) - float(g["m_cd_s1"])) < 1e-6
    vox = O.voxelize(a).numpy().reshape(3, -1)
    assert np.array_equal(np.flatnonzero(vox[0]).astype(np.int32), g["m_vox_a_idx"])
    assert np.array_equal(vox.sum(1).astype(np.int64), g["m_vox_counts"])
    for i in range(3):
        tr = O.compute_metrics(a[i], b[i])
        np.testing.assert_allclose([float(v) for v in tr], g["m_triples"][i], rtol=1e-5, atol=1e-5)
    assert abs(float(O.earth_mover_distance_sinkhorn(a, b)) - float(g["m_emd_sinkhorn_batch"])) < 1e-5


def test_set_attention(golden):
    g = golden("attention.npz")
    for C in (64, 128, 256):
        out = O.set_attention_block(sab_sd(C), "", torch.from_numpy(g[f"sab{C}_x"]), 4)
        assert rel_l2(out, g[f"sab{C}_out"]) < 2e-6, C
    eps = O.unet_attention(una_sd(), "", torch.from_numpy(g["una_x"]), torch.from_numpy(g["una_t"]))
    assert rel_l2(eps, g["una_eps"]) < 1e-5


def test_vae3d_small(golden):
    """G11: the small VAE3D (reference networks.py:1984-2206)."""
    from helpers import vae3d_small_sd
    g = golden("vae3d_small.npz")
    sd = vae3d_small_sd()
    vox = voxels_from_idx([g["occ_idx0"], g["occ_idx1"]])
    mu, logvar = O.vae3d_small_encode(sd, "", vox)
    assert rel_l2(mu, g["mu"]) < 1e-5 and rel_l2(logvar, g["logvar"]) < 1e-5
    assert rel_l2(O.vae3d_small_decode(sd, "", torch.from_numpy(g["mu"])), g["dec"]) < 1e-5


def test_linear_schedule_samplers_vs_reference(golden):
    """G16 (a2): the non-default linear schedule, whose batch-axis cumprod (diffusion.py:202) gives every
    shape of the batch its own rates.  Oracle rate tables bit-exact, sampler outputs <= 1e-5."""
    g = golden("linear.npz")
    sd = point_sd()
    model = lambda x, t: O.unet_pointnet_large(sd, "model.", x, t)
    B, T = 4, 8
    for k in range(T):
        t = torch.ones(B) - k * (1.0 / T)
        n, s = O.linear_schedule(t)
        nn_, sn = O.linear_schedule(t - 1.0 / T)
        assert np.array_equal(torch.stack([n, s, nn_, sn]).numpy(), g["sample_rates"][k])
    assert len({float(v) for v in g["sample_rates"][3, 0]}) == B          # per-shape rates really differ
    out = O.ddim_sample(model, torch.from_numpy(g["sample_xT"]), T, sched=O.linear_schedule)
    assert rel_l2(out, g["sample_out"]) < 1e-5
    out = O.ddpm_sample(model, torch.from_numpy(g["s2_xT"]), T, list(torch.from_numpy(g["s2_z"])), sched=O.linear_schedule)
    assert rel_l2(out, g["s2_out"]) < 1e-5
    noisy, nr, sr = O.add_noise(torch.from_numpy(g["s3_x0"]), torch.ones(B) * 0.3, torch.from_numpy(g["s3_noise"]),
                                sched=O.linear_schedule)
    assert np.array_equal(torch.stack([nr, sr]).numpy(), g["s3_add_rates"]) and np.array_equal(noisy.numpy(), g["s3_noisy"])
    out = O.ddim_from_state(model, noisy, torch.ones(B) * 0.3, T, sched=O.linear_schedule)
    assert rel_l2(out, g["s3_out"]) < 1e-5


def test_linear_schedule_step_tables_bit_exact(golden):
    """The product's host-side step tables (diffusion.StepTable, R = batch columns for the linear schedule)
    against the rates the reference's loops form: bit exact."""
    from shapegen_amd.diffusion import PointCloudDiffusion
    g = golden("linear.npz")
    m = PointCloudDiffusion(num_points=128, noise_schedule="linear")
    tab = m.ddim_table(8, 4)
    assert tab.width == 4 and tab.stride == 1
    got = torch.stack([tab.n, tab.s, tab.a, tab.b], dim=1).numpy()               # (T, 4, B)
    assert np.array_equal(got, g["sample_rates"])
    tab = m.ddpm_table(8, 4)
    assert np.array_equal(torch.stack([tab.n, tab.s, tab.a, tab.b], dim=1).numpy(), g["sample2_rates"])
    tab = m.from_state_table(torch.tensor(0.3), 8)
    assert tab.width == 1                                                          # 0-d t: shared by the batch
    assert np.array_equal(torch.stack([tab.n[:, 0], tab.s[:, 0]], dim=1).numpy(), g["sample3_rates"])


# ------------------------------------------------------------------ G19-G21: the 1000-step fixtures at N = 2048
def _t1000_cloud(b, n, seed):
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


def test_hash_normal_is_pinned():
    """The generator the T = 1000 DDPM fixtures rebuild their 999 per-step noise tensors from: fixed bits (additions and one
    halving in a fixed order), moments of a unit normal."""
    z = specs.hash_normal("g20.z7", 12288, 0)
    assert z.dtype == np.float64 and abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.02 and np.abs(z).max() < 6.0
    assert np.array_equal(z, specs.hash_normal("g20.z7", 12288, 0))
    assert not np.array_equal(z, specs.hash_normal("g20.z8", 12288, 0))
    u = specs.hash_uniform("g20.z7", 24, 0).reshape(2, 12)
    acc = u[:, 0].copy()
    for j in range(1, 12):
        acc += u[:, j]
    assert np.array_equal(specs.hash_normal("g20.z7", 2, 0), acc * 0.5)


def test_t1000_fixtures_last_step_with_the_oracle(golden):
    """The three 1000-step captures at N = 2048 (G19 DDIM, G20b DDPM at gain 1.0, G21 reconstruction) store the state the
    reference handed its denoiser at call 999; one oracle forward from there must land on the reference's returned cloud
    (x_0 of the last step, diffusion.py:246,283,330): the oracle pinned at the full point count at the far end of the horizon.
    G21's input is rebuilt here exactly as the GPU test rebuilds it (hashed cloud + hashed noise through `add_noise`)."""
    sd = point_sd()
    model = lambda x, t: O.unet_pointnet_large(sd, "model.", x, t)
    g = golden("point_t1000_ddim.npz")
    assert np.array_equal(g["ckpt_x"][0], g["xT"]) and int(g["ckpt_calls"][-1]) == 999
    x = torch.from_numpy(g["ckpt_x"][-1])
    t = torch.ones(2) - 999 * (1.0 / 1000)
    n, s = O.offset_cosine_schedule(t)
    assert rel_l2(O.remove_noise(x, model(x, t), n, s), g["out"]) < 2e-6
    g = golden("point_t1000_recon.npz")
    x0 = torch.from_numpy(_t1000_cloud(4, 2048, int(g["seed_cloud"])))
    eps = torch.from_numpy(specs.hash_normal("g21.eps", 4 * 2048 * 3, 0).astype(np.float32).reshape(4, 2048, 3))
    noisy, nr, sr = O.add_noise(x0, torch.ones(4) * 0.010, eps)
    assert np.array_equal(noisy.numpy(), g["noisy"])
    assert np.array_equal(np.array([nr[0].item(), sr[0].item()], np.float32), g["add_rates"])
    x = torch.from_numpy(g["ckpt_x"][-1])                       # row 0 only
    tz = torch.zeros(())
    n, s = O.offset_cosine_schedule(tz)
    assert rel_l2(O.remove_noise(x, model(x, tz.expand(1)), n, s), g["out"][:1]) < 2e-6
    cd = float(O.chamfer_distance(x0[0], torch.from_numpy(g["out"][0]), 1))
    assert abs(cd - g["cd_s1"][0]) < 1e-6


def test_t1000_ddpm_fixtures_last_step_with_the_oracle(golden):
    from helpers import as_torch
    for name, gain in (("point_t1000_ddpm_stable.npz", 1.0), ("point_t1000_ddpm.npz", 1.3)):
        g = golden(name)
        sd = as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=gain))
        x = torch.from_numpy(g["ckpt_x"][-1])
        t = torch.zeros(2)                                        # i = 0: t = 0 / T, x_t = x_0 (diffusion.py:241-257)
        n, s = O.offset_cosine_schedule(t)
        out = O.remove_noise(x, O.unet_pointnet_large(sd, "model.", x, t), n, s)
        assert rel_l2(out, g["out"]) < 2e-6, name
    assert float(np.abs(golden("point_t1000_ddpm.npz")["out"]).max()) > 1e8        # the runaway record (gain 1.3)
    assert float(np.abs(golden("point_t1000_ddpm_stable.npz")["out"]).max()) < 5e3


def test_attention_t1000_fixtures_last_step_with_the_oracle(golden):
    """G26 (`make_golden.py g26a | g26b`): the reference's `sample` / `sample2` loops over `UNetAttentionPointExperimental` at (2, 2048), 1000 steps.  One oracle
    forward from the stored call-999 state must land on the reference's returned cloud (diffusion.py:246, 283): the attention oracle pinned at the far end of the
    horizon, on the DDIM fixture (gain 1.0) and on the DDPM fixture (gain 0.6, hashed draws)."""
    from helpers import as_torch
    for name, gain, ddpm in (("attention_t1000_ddim.npz", 1.0, False), ("attention_t1000_ddpm.npz", 0.6, True)):
        g = golden(name)
        assert float(g["gain"]) == gain and int(g["ckpt_calls"][-1]) == 999 and int(g["n_draws"]) == (999 if ddpm else 0)
        tag = "g26b" if ddpm else "g26a"
        xT = specs.hash_normal(f"{tag}.xT", 2 * 2048 * 3, 0).astype(np.float32).reshape(2, 2048, 3)
        assert np.array_equal(g["ckpt_x"][0], xT)                      # the start noise the reference drew is the hashed tensor the GPU test rebuilds
        sd = as_torch(specs.synth_state_dict(specs.unet_attention_spec(), seed=0, gain=gain))
        x = torch.from_numpy(g["ckpt_x"][-1])
        t = torch.zeros(2) if ddpm else torch.ones(2) - 999 * (1.0 / 1000)
        n, s = O.offset_cosine_schedule(t)
        out = O.remove_noise(x, O.unet_attention(sd, "", x, t), n, s)
        assert np.isfinite(g["out"]).all()
        assert rel_l2(out, g["out"]) < 2e-6, name


def test_baseline_config0_as_ddpm_with_the_oracle(golden):
    """G22 (configs[0] through `sample2`: 512 points, 100 steps, batch 4, hashed per-step noise, weights at gain 1.0): the oracle's whole loop
    against the reference's returned cloud."""
    from helpers import as_torch
    g = golden("point_cfg1_ddpm.npz")
    sd = as_torch(specs.synth_state_dict(specs.unet_pointnet_large_spec(prefix="model."), seed=0, gain=float(g["gain"])))
    zs = [torch.from_numpy(specs.hash_normal(f"g22.z{k}", 4 * 512 * 3, 0).astype(np.float32).reshape(4, 512, 3)) for k in range(99)]
    out = O.ddpm_sample(lambda x, t: O.unet_pointnet_large(sd, "model.", x, t), torch.from_numpy(g["xT"]), 100, zs)
    assert rel_l2(out, g["out"]) < 1e-5


def test_latent_ddpm_1000_steps_with_the_oracle(golden):
    """G23 (`LatentDiffusion.sample2(8, num_steps=1000)`, diffusion.py:575-616, hashed per-step noise): the oracle's latent DDPM loop against
    the latent the reference handed its decoder."""
    g = golden("latent_ddpm.npz")
    sd = latent_sd()
    zs = [torch.from_numpy(specs.hash_normal(f"g23.z{k}", 8 * 256, 0).astype(np.float32).reshape(8, 256)) for k in range(999)]
    out = O.ddpm_sample(lambda z, t: O.latent_unet(sd, "model.", z, t), torch.from_numpy(g["zT"]), 1000, zs)
    assert rel_l2(out, g["z0"]) < 1e-4


def test_attention_at_2048_points_with_the_oracle(golden):
    """G24: the oracle's set-attention block and attention U-Net at BASELINE's point count against the reference's outputs."""
    g = golden("attention_n2048.npz")
    xa = torch.from_numpy(specs.hash_uniform("xa2048", 2048 * 256, 0).reshape(1, 2048, 256).astype(np.float32)) * 2
    out = O.set_attention_block(sab_sd(256), "", xa, 4)
    assert rel_l2(out[0, ::8], g["sab256_out_rows"].astype(np.float32)) < 5e-4            # the fixture is fp16
    xu = torch.from_numpy(specs.hash_uniform("xu2048", 2 * 2048 * 3, 0).reshape(2, 2048, 3).astype(np.float32)) * 1.5
    eps = O.unet_attention(una_sd(), "", xu, torch.from_numpy(g["una_t"]))
    assert rel_l2(eps, g["una_eps"]) < 1e-5
