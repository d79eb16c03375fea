"""Host-side weight folding/packing for the HIP kernels (float64 on the CPU, once per load).

Exact algebra applied to the reference's layers, none of it changes
the function being computed:
  * eval-mode BatchNorm1d folded into the preceding k=1 conv:
        W' = W * g / sqrt(var + eps),  b' = (b - mean) * g / sqrt(var + eps) + beta
  * `refine_k` (bare conv) followed by the skip half of `dec_k.conv1`:
        W_skip' = W'[:, skip] @ R_k,   b' += W'[:, skip] @ r_k
  * time channels of `enc1.conv1` and global-feature channels of `dec4.conv1`
    split off as separate matrices (they multiply vectors that are constant over N).
Weights go to fp16 [C_out][K]; biases and the tiny time path stay fp32.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

BN_EPS = 1e-5


def _np(sd, key) -> np.ndarray:
    return sd[key].detach().to("cpu", torch.float64).numpy()


def fold_conv_bn(sd, conv: str, bn: str | None) -> Tuple[np.ndarray, np.ndarray]:
    w = _np(sd, conv + ".weight")
    w = w.reshape(w.shape[0], -1)
    b = _np(sd, conv + ".bias")
    if bn is not None:
        scale = _np(sd, bn + ".weight") / np.sqrt(_np(sd, bn + ".running_var") + BN_EPS)
        w = w * scale[:, None]
        b = (b - _np(sd, bn + ".running_mean")) * scale + _np(sd, bn + ".bias")
    return w, b


def pack_point_unet(sd: Dict[str, torch.Tensor], prefix: str, time_dim: int, dim: int):
    """Returns (lin, extras): lin = list of 26 (W fp16-able float64 [C][K], b float64 [C]) in the
    execution order documented in csrc/unet.hip; extras = dict of fp32-able arrays."""
    if dim != time_dim:
        # same failure the reference hits at enc1.conv1 (expects 3+time_dim channels, gets 3+dim)
        raise RuntimeError(f"UNetPointNetLarge needs dim == time_dim (got dim={dim}, time_dim={time_dim}): "
                           "enc1 is built for 3+time_dim input channels (reference networks.py:744,797)")
    p = prefix
    lin: List[Tuple[np.ndarray, np.ndarray]] = []
    ex: Dict[str, np.ndarray] = {}
    ex["tw0"], ex["tb0"] = _np(sd, p + "time_mlp.0.weight"), _np(sd, p + "time_mlp.0.bias")
    ex["tw2"], ex["tb2"] = _np(sd, p + "time_mlp.2.weight"), _np(sd, p + "time_mlp.2.bias")
    w, b = fold_conv_bn(sd, p + "enc1.conv1", p + "enc1.bn1")
    ex["e1w_xyz"], ex["e1w_t"], ex["e1b"] = w[:, :3].copy(), w[:, 3:].copy(), b
    lin.append(fold_conv_bn(sd, p + "enc1.conv2", p + "enc1.bn2"))
    lin.append(fold_conv_bn(sd, p + "enc1.conv3", p + "enc1.bn3"))
    for name in ("enc2", "enc3", "enc4"):
        for i in (1, 2, 3):
            lin.append(fold_conv_bn(sd, f"{p}{name}.conv{i}", f"{p}{name}.bn{i}"))
    lin.append(fold_conv_bn(sd, p + "global_feat.0", p + "global_feat.1"))
    lin.append(fold_conv_bn(sd, p + "global_feat.3", p + "global_feat.4"))
    for name, k, split in (("dec4", 4, 4096), ("dec3", 3, 512), ("dec2", 2, 256), ("dec1", 1, 128)):
        w, b = fold_conv_bn(sd, f"{p}{name}.conv1", f"{p}{name}.bn1")
        r_w = _np(sd, f"{p}refine{k}.weight")
        r_w = r_w.reshape(r_w.shape[0], -1)
        r_b = _np(sd, f"{p}refine{k}.bias")
        w_prev, w_skip = w[:, :split], w[:, split:]
        b = b + w_skip @ r_b
        w_skip = w_skip @ r_w
        if name == "dec4":
            ex["wg"] = w_prev.copy()          # [1024][4096], multiplies the pooled global feature
            lin.append((w_skip, b))           # bias becomes part of the per-shape bias
        else:
            lin.append((np.concatenate([w_prev, w_skip], axis=1), b))
        lin.append(fold_conv_bn(sd, f"{p}{name}.conv2", f"{p}{name}.bn2"))
        lin.append(fold_conv_bn(sd, f"{p}{name}.conv3", f"{p}{name}.bn3"))
    lin.append(fold_conv_bn(sd, p + "output.0", p + "output.1"))
    hw, hb = fold_conv_bn(sd, p + "output.3", None)
    ex["head_w"], ex["head_b"] = hw, hb
    assert len(lin) == 26
    return lin, ex


def split_hilo(w: np.ndarray) -> np.ndarray:
    """[C][K] float64 weights -> fp16 [C][2 K] = the fp16 weights | the fp16 of their rounding residuals (hi / lo weights: the narrow layers
    of the point denoiser, `csrc/chain.hip`, `pcd_gemm_f16_hilo`).  hi + lo reproduces w to ~2^-22 relative: the kernels run their K loop twice."""
    w = np.asarray(w, np.float64)
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float64)).astype(np.float16)
    return np.concatenate([hi, lo], axis=1)


def timestep_freqs(time_dim: int) -> torch.Tensor:
    """f_j of the sinusoidal embedding, computed with the reference's exact torch ops
    (networks.py:831-833) so the fp32 table is bit-identical."""
    half = time_dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half - 1)
    return torch.exp(torch.arange(half) * -step).float()


def pack_latent_unet(sd: Dict[str, torch.Tensor], prefix: str):
    """SimpleLatentUNetPointNet (dim=512, latent=256, time=256): returns (lin, gn, extras) in the
    execution order of csrc/latent.hip.  refine_k is folded into the skip half of dec_k; the time
    half of enc1 becomes `e1w_t` (a per-t bias computed by pcd_time_embed)."""
    p = prefix
    g = lambda k: _np(sd, p + k)
    ex = {"tw0": g("time_mlp.0.weight"), "tb0": g("time_mlp.0.bias"),
          "tw2": g("time_mlp.2.weight"), "tb2": g("time_mlp.2.bias")}
    lin, gn = [], []
    w, b = g("enc1.0.weight"), g("enc1.0.bias")
    latent = w.shape[1] - ex["tw2"].shape[0]
    ex["e1w_t"], ex["e1b"] = w[:, latent:].copy(), b
    lin.append((w[:, :latent].copy(), b)); gn.append((g("enc1.1.weight"), g("enc1.1.bias")))
    for name in ("enc2", "enc3", "enc4"):
        lin.append((g(name + ".0.weight"), g(name + ".0.bias"))); gn.append((g(name + ".1.weight"), g(name + ".1.bias")))
    lin.append((g("global_feat.0.weight"), g("global_feat.0.bias"))); gn.append((g("global_feat.1.weight"), g("global_feat.1.bias")))
    lin.append((g("global_feat.3.weight"), g("global_feat.3.bias"))); gn.append((g("global_feat.4.weight"), g("global_feat.4.bias")))
    for name, k in (("dec4", 4), ("dec3", 3), ("dec2", 2), ("dec1", 1)):
        w, b = g(name + ".0.weight"), g(name + ".0.bias")
        r_w, r_b = g(f"refine{k}.weight"), g(f"refine{k}.bias")
        split = w.shape[1] - r_w.shape[0]
        b = b + w[:, split:] @ r_b
        w = np.concatenate([w[:, :split], w[:, split:] @ r_w], axis=1)
        lin.append((w, b)); gn.append((g(name + ".1.weight"), g(name + ".1.bias")))
    lin.append((g("output.0.weight"), g("output.0.bias")))
    lin.append((g("output.2.weight"), g("output.2.bias")))
    return lin, gn, ex
