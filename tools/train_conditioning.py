"""Conditioning of the point denoiser's training step: how much the ORACLE's own gradients move under a
tiny input perturbation, and how far its fp32 prediction is from its fp64 one.  CPU only (PyTorch autograd on the
oracle's functional restatement); the numbers justify the loose whole-step bounds of tests/test_gpu_train.py.

    python tools/train_conditioning.py            # ~1 minute on 8 cores
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import shapegen_amd  # noqa: E402,F401
from helpers import point_sd  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)
    B, N = 4, 256
    x = torch.randn(B, N, 3, generator=g) * 0.5
    t = torch.rand(B, generator=g)
    noise = torch.randn(B, N, 3, generator=g)

    def step(xin, dtype=torch.float32):
        sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in point_sd().items()}
        loss, grads = O.point_training_step(sd, "model.", xin.to(dtype), t.to(dtype), noise.to(dtype))
        return float(loss), {k: v.double() for k, v in grads.items()}

    l0, g0 = step(x)
    for eps in (1e-5, 1e-4, 3e-4, 1e-3):
        l1, g1 = step(x + eps * torch.randn(x.shape, generator=g))
        num = sum(float((g1[k] - g0[k]).pow(2).sum()) for k in g0) ** 0.5
        den = sum(float(g0[k].pow(2).sum()) for k in g0) ** 0.5
        cos = [float((g0[k] * g1[k]).sum() / (g0[k].norm() * g1[k].norm() + 1e-30)) for k in g0 if g0[k].numel() > 8]
        print(f"input perturbation {eps:7.0e}: loss {l0:.5f} -> {l1:.5f}; gradient change rel-L2 {num / den:.3f}; "
              f"per-tensor cosine min {min(cos):.3f} median {sorted(cos)[len(cos) // 2]:.3f}")
    l64, g64 = step(x, torch.float64)
    num = sum(float((g64[k] - g0[k]).pow(2).sum()) for k in g0) ** 0.5
    den = sum(float(g64[k].pow(2).sum()) for k in g0) ** 0.5
    print(f"fp32 v. fp64 oracle on identical inputs: loss {l0:.6f} v. {l64:.6f}; gradient rel-L2 {num / den:.3e}")


if __name__ == "__main__":
    main()
