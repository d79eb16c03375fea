"""VAE3DLarge training step on the HIP kernels (shapegen_amd.training_vae.VAETrainer) against the oracle's autograd
(oracle.torch_oracle.vae_training_step, pinned to the reference by tests/golden/train_vae.npz).  With fp16 operands
through ~30 conv layers the end-to-end gradient comparison is a direction / magnitude check (measured: cosine >= 0.99,
norm ratio 0.99..1.02 on every weight tensor); the layer arithmetic is pinned exactly by the im2col / col2im / BatchNorm
kernel tests in test_gpu_train.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2, as_torch
from oracle import torch_oracle as O
from oracle import make_golden as MG
from shapegen_amd import specs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    with torch.enable_grad():
        yield


def _vae_sd():
    return as_torch(specs.synth_state_dict(specs.vae3d_large_spec(prefix="vae."), seed=0, gain=1.3))


def test_vae_training_step_against_oracle():
    from shapegen_amd.training_vae import VAETrainer
    from shapegen_amd.vae import VAE3DLarge
    sd = _vae_sd()
    vae = VAE3DLarge()
    vae.load_state_dict({k[len("vae."):]: v for k, v in sd.items()}, strict=True)
    vae = vae.to("cuda")
    x = torch.from_numpy(MG.synth_voxels(2, 5))
    eps = torch.randn(2, 256, generator=torch.Generator().manual_seed(1))
    tr = VAETrainer(vae, lr=1e-4)
    tr.forward(x.cuda(), eps.cuda())
    loss, recon_loss, kl = tr.backward(0.01)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    l_ref, r_ref, k_ref, recon_ref, mu_ref, lv_ref, grads_ref = O.vae_training_step(sd_ref, "vae.", x, eps, 0.01, specs.VAE_ENC, specs.VAE_DEC)
    # measured: mu / logvar 2e-3, loss 3e-4, reconstruction 1.5e-3, gradient cosines min 0.992 / median 0.997
    assert rel_l2(tr.mu.cpu(), mu_ref) < 1e-2 and rel_l2(tr.logvar.cpu(), lv_ref) < 1e-2
    assert abs(kl.item() - k_ref.item()) < 1e-2 * abs(k_ref.item())
    assert abs(recon_loss.item() - r_ref.item()) < 5e-3 * r_ref.item() and abs(loss.item() - l_ref.item()) < 5e-3 * l_ref.item()
    assert rel_l2(tr.recon.cpu(), recon_ref) < 1e-2
    grads = tr.grads()
    cos = {}
    for k, gr in grads_ref.items():
        mine = grads[k[len("vae."):]].cpu()
        assert mine.shape == gr.shape and torch.isfinite(mine).all(), k
        if gr.dim() > 1 and gr.norm() > 0:
            cos[k] = F.cosine_similarity(mine.reshape(1, -1), gr.reshape(1, -1)).item()
            assert 0.9 < mine.norm().item() / gr.norm().item() < 1.1, (k, mine.norm().item(), gr.norm().item())
    low = sorted(cos.items(), key=lambda kv: kv[1])[:5]
    assert min(cos.values()) > 0.95 and np.median(list(cos.values())) > 0.98, low
    for k, v in vae.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            assert torch.allclose(v.cpu(), sd_ref["vae." + k], rtol=3e-2, atol=3e-2), k


def test_vae_training_reduces_the_loss():
    from shapegen_amd.training_vae import VAETrainer
    from shapegen_amd.vae import VAE3DLarge
    torch.manual_seed(0)
    vae = VAE3DLarge().to("cuda")                       # the reference's own initialisation
    tr = VAETrainer(vae, lr=1e-3)
    x = torch.from_numpy(MG.synth_voxels(4, 7)).cuda()
    eps = torch.randn(4, 256, device="cuda")
    losses = [float(tr.train_step(x, 0.01, eps)[0]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < 0.85 * losses[0], losses
    vae.eval()
    rec, mu, logvar = vae(x)
    assert torch.isfinite(rec).all() and rec.shape == x.shape
