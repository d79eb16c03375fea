"""Lightning-free checkpoint loading: a `.ckpt` in the reference's layout
(`state_dict` + `hyper_parameters`, the latter pickled as Lightning's AttributeDict) loads into the
drop-in classes without Lightning installed."""
import sys
import types

import pytest
import torch

from helpers import point_sd, latent_sd


def _fake_lightning_attrdict():
    """Pickle hyper_parameters as `pytorch_lightning.utilities.parsing.AttributeDict`, as Lightning does."""
    pkg = types.ModuleType("pytorch_lightning")
    util = types.ModuleType("pytorch_lightning.utilities")
    parsing = types.ModuleType("pytorch_lightning.utilities.parsing")

    class AttributeDict(dict):
        pass

    AttributeDict.__module__ = "pytorch_lightning.utilities.parsing"
    AttributeDict.__qualname__ = "AttributeDict"
    parsing.AttributeDict = AttributeDict
    sys.modules.update({"pytorch_lightning": pkg, "pytorch_lightning.utilities": util,
                        "pytorch_lightning.utilities.parsing": parsing})
    return AttributeDict


def _drop_fake_lightning():
    for k in ("pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"):
        sys.modules.pop(k, None)


def test_point_checkpoint_roundtrip(tmp_path):
    from shapegen_amd.diffusion import PointCloudDiffusion
    AD = _fake_lightning_attrdict()
    sd = point_sd()
    path = tmp_path / "point_cloud_diffusion-epoch=47-val_loss=0.16.ckpt"
    try:
        torch.save({"state_dict": sd, "epoch": 47, "optimizer_states": [],
                    "hyper_parameters": AD(num_points=2048, dim=256, time_dim=256, lr=1e-4, noise_schedule="cosine")},
                   path)
    finally:
        _drop_fake_lightning()                      # loading must work WITHOUT Lightning importable
    m = PointCloudDiffusion.load_from_checkpoint(str(path))
    assert m.hparams.num_points == 2048 and m.noise_schedule == "cosine"
    got = m.state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError):
        PointCloudDiffusion.load_from_checkpoint(str(_plain(tmp_path)))


def _plain(tmp_path):
    p = tmp_path / "not_lightning.ckpt"
    torch.save({"weights": 1}, p)
    return p


def test_latent_checkpoint_needs_vae(tmp_path):
    from shapegen_amd.diffusion import LatentDiffusion
    from shapegen_amd.vae import VAE3DLarge
    sd = latent_sd()
    path = tmp_path / "ldm.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"latent_dim": 256, "dim": 512, "time_dim": 256, "lr": 1e-4,
                                                       "noise_schedule": "cosine", "is_voxel_based": True}}, path)
    with pytest.raises(TypeError):
        LatentDiffusion.load_from_checkpoint(str(path))          # hyper-parameters are saved with ignore=['vae']
    m = LatentDiffusion.load_from_checkpoint(str(path), vae=VAE3DLarge())
    got = m.state_dict()
    assert set(got.keys()) == set(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert not any(p.requires_grad for p in m.vae.parameters())  # frozen VAE, diffusion.py:377-378


def test_latent_diffusion_construction_reinitialises_vae_linear_heads():
    """SURVEY a16 / reference diffusion.py:392-408: `init_weights` skips the child named 'vae' but walks
    `self.modules()`, so constructing a LatentDiffusion re-initialises every nn.Linear of the VAE in place
    (kaiming-normal fan_out weights, zero bias) and leaves Conv3d / BatchNorm3d alone.  Restated bug for bug."""
    import numpy as np
    from shapegen_amd import specs
    from shapegen_amd.diffusion import LatentDiffusion
    from shapegen_amd.vae import VAE3D, VAE3DLarge
    for cls, spec, heads in ((VAE3DLarge, specs.vae3d_large_spec(), ("fc_mu", "fc_logvar", "decoder_input")),
                             (VAE3D, specs.vae3d_small_spec(), ("encoder.5", "fc_mu", "fc_logvar", "decoder_input"))):
        vae = cls()
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in specs.synth_state_dict(spec, seed=3, gain=1.3).items()}
        vae.load_state_dict(sd, strict=True)
        torch.manual_seed(5)
        LatentDiffusion(vae)
        got = vae.state_dict()
        for k, v in sd.items():
            head = any(k.startswith(h + ".") for h in heads)
            if not head:
                assert torch.equal(got[k], v), k                       # convs, norms, running statistics untouched
            elif k.endswith(".bias"):
                assert torch.count_nonzero(got[k]) == 0, k
            else:
                assert not torch.equal(got[k], v), k
                std = float(got[k].std())
                assert abs(std / (2.0 / v.shape[0]) ** 0.5 - 1) < 0.05, (k, std)   # kaiming fan_out: std = sqrt(2 / out_features)
        assert not any(p.requires_grad for p in vae.parameters())


def test_parent_load_state_dict_invalidates_child_packed_weights():
    """A parent's load_state_dict never calls a child's override; the post hook must drop the packed device weights."""
    from shapegen_amd.diffusion import PointCloudDiffusion
    m = PointCloudDiffusion(num_points=64)
    m.model._packed = {"stale": True}
    m.load_state_dict(m.state_dict(), strict=True)
    assert m.model._packed is None
