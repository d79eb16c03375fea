"""Pins the training-step restatement in oracle/torch_oracle.py (`point_training_step`, `adamw_step`) against
tests/golden/train.npz, captured from the reference's PointCloudDiffusion in train() mode with autograd and
torch.optim.AdamW (`python oracle/make_golden.py train`).  Same ATen ops => tight tolerances."""
import numpy as np
import torch

from helpers import point_sd
from oracle import torch_oracle as O
from shapegen_amd import specs


def _digest_idx(name, numel):
    return (np.abs(specs.hash_uniform("digest." + name, 64, 7)) * (numel - 1)).astype(np.int64)


def test_training_step_matches_reference(golden):
    g = golden("train.npz")
    sd = point_sd()
    nbt0 = {k: int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")}
    x_t, t, noise = (torch.from_numpy(g[k]) for k in ("x_t", "t", "noise"))
    xt2 = O.add_noise(torch.from_numpy(g["x0"]), t, noise)[0]
    assert torch.equal(xt2, x_t)
    loss, grads = O.point_training_step(sd, "model.", x_t, t, noise)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    names = [str(n) for n in g["param_names"]]
    assert sorted(names) == sorted(grads.keys())
    for k in names:
        flat = grads[k].reshape(-1).double()
        want = g["grad." + k]
        got = np.concatenate([[flat.norm().item(), flat.sum().item()], flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy()])
        scale = max(want[0], 1e-12)
        assert np.abs(got[0] - want[0]) <= 1e-4 * scale, k
        assert np.abs(got[2:] - want[2:]).max() <= 1e-4 * max(np.abs(want[2:]).max(), 1e-3 * scale / flat.numel() ** 0.5) + 1e-9, k
    # BatchNorm running statistics after the step
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var")):
            assert np.allclose(v.numpy(), g["buf1." + k], rtol=1e-5, atol=1e-6), k
        if k.endswith("num_batches_tracked"):
            assert int(v) == nbt0[k] + 1
    # one AdamW update (diffusion.py:60)
    params = {k: sd[k] for k in names}
    O.adamw_step(params, grads, {}, lr=1e-4, weight_decay=1e-5)
    for k in names:
        flat = params[k].reshape(-1).double()
        got = flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy()
        assert np.abs(got - g["param1." + k]).max() <= 2e-6, k


def test_latent_training_step_matches_reference(golden):
    """oracle.latent_training_step (train() mode: the recorded Dropout keep mask) against tests/golden/train_latent.npz."""
    from helpers import latent_sd
    g = golden("train_latent.npz")
    sd = {k: v for k, v in latent_sd().items() if k.startswith("model.")}
    z_t, t, noise, mask = (torch.from_numpy(g[k]) for k in ("z_t", "t", "noise", "mask"))
    loss, pred, grads = O.latent_training_step(sd, "model.", z_t, t, noise, mask)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 and np.abs(pred.numpy() - g["pred"]).max() <= 1e-5
    names = [str(n) for n in g["param_names"]]
    assert sorted("model." + n for n in names) == sorted(grads.keys())
    params = {}
    for k in names:
        flat = grads["model." + k].reshape(-1).double()
        want = g["grad." + k]
        got = np.concatenate([[flat.norm().item(), flat.sum().item()], flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy()])
        assert abs(got[0] - want[0]) <= 1e-4 * max(want[0], 1e-12), k
        assert np.abs(got[2:] - want[2:]).max() <= 1e-4 * max(np.abs(want[2:]).max(), 1e-9) + 1e-9, k
        params["model." + k] = sd["model." + k].clone()
    O.adamw_step(params, grads, {}, lr=1e-4, weight_decay=1e-5)
    for k in names:
        flat = params["model." + k].reshape(-1).double()
        assert np.abs(flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy() - g["param1." + k]).max() <= 2e-6, k


def test_vae_training_step_matches_reference(golden):
    """oracle.vae_training_step / vae_kl_weight against tests/golden/train_vae.npz (VAE3DLarge.calculate_loss in train()
    mode, autograd, one torch.optim.Adam step, the KL warm-up / annealing schedule)."""
    from helpers import as_torch
    from oracle import make_golden as MG
    g = golden("train_vae.npz")
    sd = as_torch(specs.synth_state_dict(specs.vae3d_large_spec(prefix="vae."), seed=0, gain=1.3))
    x = torch.from_numpy(MG.synth_voxels(2, 5))
    eps = torch.from_numpy(g["eps"])
    for e, mx, want in g["kl_weights"]:
        assert abs(O.vae_kl_weight(int(e), int(mx)) - want) < 1e-12
    w = O.vae_kl_weight(0, 100)
    loss, rl, kl, recon, mu, lv, grads = O.vae_training_step(sd, "vae.", x, eps, w, specs.VAE_ENC, specs.VAE_DEC)
    assert abs(loss.item() - float(g["loss"])) <= 2e-6
    assert np.abs(recon.reshape(-1)[::997].numpy() - g["recon_sample"]).max() <= 1e-5
    names = [str(n) for n in g["param_names"]]
    assert sorted("vae." + n for n in names) == sorted(grads.keys())
    params = {}
    for k in names:
        flat = grads["vae." + k].reshape(-1).double()
        want = g["grad." + k]
        got = np.concatenate([[flat.norm().item(), flat.sum().item()], flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy()])
        assert abs(got[0] - want[0]) <= 2e-4 * max(want[0], 1e-12), k
        assert np.abs(got[2:] - want[2:]).max() <= 2e-4 * max(np.abs(want[2:]).max(), 1e-3 * want[0] / flat.numel() ** 0.5) + 1e-10, k
        params["vae." + k] = sd["vae." + k].clone()
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var")):
            assert np.allclose(v.numpy(), g["buf1." + k[len("vae."):]], rtol=1e-4, atol=1e-6), k
    O.adamw_step(params, grads, {}, lr=1e-4, weight_decay=0.0)             # Adam = AdamW without decay
    for k in names:
        flat = params["vae." + k].reshape(-1).double()
        got = flat[torch.from_numpy(_digest_idx(k, flat.numel()))].numpy()
        # the first Adam step is lr * g / (|g| + eps): elements with |g| ~ 1e-8 amplify last-bit gradient differences
        assert np.abs(got - g["param1." + k]).max() <= 2e-5, k
