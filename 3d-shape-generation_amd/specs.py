"""Architecture tables and parameter specs for the sampler hot path.

Everything here is *data*: channel tables that describe the reference's
networks (reference `networks.py:724-777` UNetPointNetLarge, `:962-1049`
SimpleLatentUNetPointNet, `:2208-2264` VAE3DLarge, `:51-68` SetAttentionBlock,
`:597-650` UNetAttentionPointExperimental) and, derived from them, the ordered
`state_dict` key/shape list that is the weight contract.

`synth_state_dict` is a deterministic, libm-free weight generator (integer hash
-> uniform floats) so that the container that captures golden vectors and the
GPU box regenerate bit-identical weights without shipping 86 MB of fixtures.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

Spec = List[Tuple[str, Tuple[int, ...], str]]  # (key, shape, role)

# roles: w (conv/linear weight), b (bias), g (norm scale), beta (norm shift),
#        rm (running mean), rv (running var), nbt (num_batches_tracked)


def _conv1d(spec: Spec, p: str, cin: int, cout: int) -> None:
    spec.append((p + ".weight", (cout, cin, 1), "w"))
    spec.append((p + ".bias", (cout,), "b"))


def _linear(spec: Spec, p: str, cin: int, cout: int) -> None:
    spec.append((p + ".weight", (cout, cin), "w"))
    spec.append((p + ".bias", (cout,), "b"))


def _bn(spec: Spec, p: str, c: int) -> None:
    spec.append((p + ".weight", (c,), "g"))
    spec.append((p + ".bias", (c,), "beta"))
    spec.append((p + ".running_mean", (c,), "rm"))
    spec.append((p + ".running_var", (c,), "rv"))
    spec.append((p + ".num_batches_tracked", (), "nbt"))


def _affine(spec: Spec, p: str, c: int) -> None:  # GroupNorm / LayerNorm
    spec.append((p + ".weight", (c,), "g"))
    spec.append((p + ".bias", (c,), "beta"))


def _pointnet_layer(spec: Spec, p: str, cin: int, mid: int, cout: int) -> None:
    # reference networks.py:16-49: three (conv k=1 -> BN -> ReLU) stages
    chans = [(cin, mid), (mid, mid), (mid, cout)]
    for i, (a, b) in enumerate(chans, start=1):
        _conv1d(spec, f"{p}.conv{i}", a, b)
        _bn(spec, f"{p}.bn{i}", b)


# ---------------------------------------------------------------- point UNet
# (name, c_in, c_mid, c_out) of the PointNetLayers, in registration order.
POINT_ENC = [("enc1", None, 64, 128), ("enc2", 128, 128, 256),
             ("enc3", 256, 256, 512), ("enc4", 512, 512, 1024)]
POINT_GLOBAL = [(1024, 2048), (2048, 4096)]
POINT_DEC = [("dec4", 4096 + 1024, 1024, 512), ("dec3", 512 + 512, 512, 256),
             ("dec2", 256 + 256, 256, 128), ("dec1", 128 + 128, 128, 64)]
POINT_REFINE = [("refine1", 128), ("refine2", 256), ("refine3", 512), ("refine4", 1024)]


def unet_pointnet_large_spec(dim: int = 256, time_dim: int = 256, prefix: str = "") -> Spec:
    s: Spec = []
    _linear(s, prefix + "time_mlp.0", time_dim, dim)
    _linear(s, prefix + "time_mlp.2", dim, dim)
    for name, cin, mid, cout in POINT_ENC:
        _pointnet_layer(s, prefix + name, 3 + time_dim if cin is None else cin, mid, cout)
    (a0, b0), (a1, b1) = POINT_GLOBAL
    _conv1d(s, prefix + "global_feat.0", a0, b0)
    _bn(s, prefix + "global_feat.1", b0)
    _conv1d(s, prefix + "global_feat.3", a1, b1)
    _bn(s, prefix + "global_feat.4", b1)
    for name, cin, mid, cout in POINT_DEC:
        _pointnet_layer(s, prefix + name, cin, mid, cout)
    _conv1d(s, prefix + "output.0", 64, 64)
    _bn(s, prefix + "output.1", 64)
    _conv1d(s, prefix + "output.3", 64, 3)
    for name, c in POINT_REFINE:
        _conv1d(s, prefix + name, c, c)
    return s


# --------------------------------------------------------------- latent UNet
def latent_unet_spec(latent_dim: int = 256, dim: int = 512, time_dim: int = 256,
                     prefix: str = "") -> Spec:
    d = dim
    s: Spec = []
    _linear(s, prefix + "time_mlp.0", time_dim, time_dim)
    _linear(s, prefix + "time_mlp.2", time_dim, time_dim)
    enc = [("enc1", latent_dim + time_dim, d // 4), ("enc2", d // 4, d // 2),
           ("enc3", d // 2, d), ("enc4", d, d * 2)]
    for name, a, b in enc:
        _linear(s, f"{prefix}{name}.0", a, b)
        _affine(s, f"{prefix}{name}.1", b)
    _linear(s, prefix + "global_feat.0", d * 2, d * 4)
    _affine(s, prefix + "global_feat.1", d * 4)
    _linear(s, prefix + "global_feat.3", d * 4, d * 8)
    _affine(s, prefix + "global_feat.4", d * 8)
    dec = [("dec4", d * 8 + d * 2, d * 2), ("dec3", d * 2 + d, d),
           ("dec2", d + d // 2, d // 2), ("dec1", d // 2 + d // 4, d // 4)]
    for name, a, b in dec:
        _linear(s, f"{prefix}{name}.0", a, b)
        _affine(s, f"{prefix}{name}.1", b)
    _linear(s, prefix + "output.0", d // 4, d // 4)
    _linear(s, prefix + "output.2", d // 4, latent_dim)
    for name, c in [("refine1", d // 4), ("refine2", d // 2), ("refine3", d), ("refine4", d * 2)]:
        _linear(s, prefix + name, c, c)
    return s


# ----------------------------------------------------------------- voxel VAE
def _conv3d(spec: Spec, p: str, cin: int, cout: int, k: int) -> None:
    spec.append((p + ".weight", (cout, cin, k, k, k), "w"))
    spec.append((p + ".bias", (cout,), "b"))


def _convT3d(spec: Spec, p: str, cin: int, cout: int, k: int) -> None:
    spec.append((p + ".weight", (cin, cout, k, k, k), "wT"))
    spec.append((p + ".bias", (cout,), "b"))


def _res3d(spec: Spec, p: str, cin: int, cout: int) -> None:
    # reference networks.py:471-504
    _conv3d(spec, p + ".conv1", cin, cout, 3)
    _bn(spec, p + ".bn1", cout)
    _conv3d(spec, p + ".conv2", cout, cout, 3)
    _bn(spec, p + ".bn2", cout)
    if cin != cout:
        _conv3d(spec, p + ".downsample", cin, cout, 1)


# Encoder/decoder programs: (sequential index, op, args). ReLU/Flatten/Sigmoid
# indices are skipped in the key numbering exactly as nn.Sequential would.
VAE_ENC = [(0, "conv", (1, 32, 3, 1, 1)), (2, "res", (32, 64)),
           (3, "conv", (64, 64, 4, 2, 1)), (5, "res", (64, 128)),
           (6, "conv", (128, 128, 4, 2, 1)), (8, "res", (128, 256)),
           (9, "conv", (256, 256, 4, 2, 1)), (11, "res", (256, 512)),
           (12, "conv", (512, 512, 4, 1, 0))]
VAE_DEC = [(0, "convT", (512, 256, 4, 2, 1)), (2, "res", (256, 256)),
           (3, "convT", (256, 128, 4, 2, 1)), (5, "res", (128, 128)),
           (6, "convT", (128, 64, 4, 2, 1)), (8, "res", (64, 64)),
           (9, "conv", (64, 32, 3, 1, 1)), (11, "res", (32, 32)),
           (12, "conv", (32, 1, 3, 1, 1))]


def vae3d_large_spec(latent_dim: int = 256, prefix: str = "") -> Spec:
    s: Spec = []
    for idx, op, a in VAE_ENC:
        if op == "conv":
            _conv3d(s, f"{prefix}encoder.{idx}", a[0], a[1], a[2])
        else:
            _res3d(s, f"{prefix}encoder.{idx}", a[0], a[1])
    _linear(s, prefix + "fc_mu", 512, latent_dim)
    _linear(s, prefix + "fc_logvar", 512, latent_dim)
    _linear(s, prefix + "decoder_input", latent_dim, 512 * 4 * 4 * 4)
    for idx, op, a in VAE_DEC:
        if op == "conv":
            _conv3d(s, f"{prefix}decoder.{idx}", a[0], a[1], a[2])
        elif op == "convT":
            _convT3d(s, f"{prefix}decoder.{idx}", a[0], a[1], a[2])
        else:
            _res3d(s, f"{prefix}decoder.{idx}", a[0], a[1])
    return s


def vae3d_small_spec(latent_dim: int = 256, prefix: str = "") -> Spec:
    """VAE3D (reference networks.py:1984-2021): stride-2 Conv3DBlocks / Deconv3DBlocks (k3)."""
    s: Spec = []
    for i, (a, b) in enumerate([(1, 32), (32, 64), (64, 128), (128, 256)]):
        _conv3d(s, f"{prefix}encoder.{i}.conv", a, b, 3)
        _bn(s, f"{prefix}encoder.{i}.bn", b)
    _linear(s, prefix + "encoder.5", 256 * 8, 512)
    _linear(s, prefix + "fc_mu", 512, latent_dim)
    _linear(s, prefix + "fc_logvar", 512, latent_dim)
    _linear(s, prefix + "decoder_input", latent_dim, 256 * 8)
    for i, (a, b) in enumerate([(256, 128), (128, 64), (64, 32)]):
        _convT3d(s, f"{prefix}decoder.{i}.deconv", a, b, 3)
        _bn(s, f"{prefix}decoder.{i}.bn", b)
    _convT3d(s, prefix + "decoder.3", 32, 1, 3)
    return s


# ------------------------------------------------------------- set attention
def set_attention_spec(dim: int, prefix: str = "") -> Spec:
    s: Spec = []
    s.append((prefix + "attention.in_proj_weight", (3 * dim, dim), "w"))
    s.append((prefix + "attention.in_proj_bias", (3 * dim,), "b"))
    _linear(s, prefix + "attention.out_proj", dim, dim)
    _affine(s, prefix + "ln1", dim)
    _linear(s, prefix + "ff.0", dim, 4 * dim)
    _linear(s, prefix + "ff.2", 4 * dim, dim)
    _affine(s, prefix + "ln2", dim)
    return s


ATTN_UNET_EMB = [("emb1", 3), ("emb2", 64), ("emb3", 128),
                 ("emb_dec3", 256), ("emb_dec2", 128), ("emb_dec1", 64)]


def unet_attention_spec(dim: int = 256, time_dim: int = 256, prefix: str = "") -> Spec:
    s: Spec = []
    for name, c in ATTN_UNET_EMB:
        _linear(s, prefix + name, time_dim, c)
    _linear(s, prefix + "time_mlp.0", time_dim, dim)
    _linear(s, prefix + "time_mlp.2", dim, dim)
    _pointnet_layer(s, prefix + "enc1", 3, 64, 64)
    s += set_attention_spec(64, prefix + "att1.")
    _pointnet_layer(s, prefix + "enc2", 64, 128, 128)
    s += set_attention_spec(128, prefix + "att2.")
    _pointnet_layer(s, prefix + "enc3", 128, 256, 256)
    s += set_attention_spec(256, prefix + "att3.")
    s += set_attention_spec(256, prefix + "bottleneck.")
    s += set_attention_spec(256, prefix + "att_dec3.")
    _pointnet_layer(s, prefix + "dec3", 512, 128, 128)
    s += set_attention_spec(128, prefix + "att_dec2.")
    _pointnet_layer(s, prefix + "dec2", 256, 64, 64)
    s += set_attention_spec(64, prefix + "att_dec1.")
    _pointnet_layer(s, prefix + "dec1", 128, 3, 3)
    _conv1d(s, prefix + "output", 3, 3)
    return s


# -------------------------------------------------- deterministic generator
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (name, seed, index)."""
    key = np.uint64(zlib.crc32(name.encode("utf-8")) + (int(seed) << 32))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(idx ^ key) + key)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def hash_normal(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values, approximately N(0, 1): half the sum of twelve `hash_uniform` draws (Irwin-Hall; variance
    12 / 3 / 4 = 1).  Elementwise additions in a fixed order and one exact halving only, so every platform rebuilds the same bits: this is how the
    T = 1000 DDPM fixtures regenerate their 1000 per-step noise tensors instead of storing them."""
    u = hash_uniform(name, 12 * n, seed).reshape(n, 12)
    acc = u[:, 0].copy()
    for j in range(1, 12):          # fixed left-to-right order (a library reduction may pair the terms differently)
        acc += u[:, j]
    return acc * 0.5


def _fan_in(shape: Tuple[int, ...], role: str) -> int:
    if role == "wT":  # ConvTranspose3d (cin, cout, k,k,k): each output sees cin*k^3/stride^3
        return shape[0] * int(np.prod(shape[2:])) // 8
    return int(np.prod(shape[1:]))


def synth_state_dict(spec: Spec, seed: int = 0, gain: float = 1.3,
                     overrides: Dict[str, float] | None = None) -> Dict[str, np.ndarray]:
    """Deterministic 'tamed' weights: fan-in scaled uniforms, non-trivial norm stats.

    `overrides` maps a key substring to a multiplicative factor on that tensor
    (used to keep the predicted noise O(1) over long sampling horizons).
    """
    out: Dict[str, np.ndarray] = {}
    for key, shape, role in spec:
        n = int(np.prod(shape)) if shape else 1
        u = hash_uniform(key, n, seed)
        if role in ("w", "wT"):
            std = gain / np.sqrt(max(_fan_in(shape, role), 1))
            v = u * (np.sqrt(3.0) * std)
        elif role == "b":
            v = u * 0.05
        elif role == "g":
            v = 1.0 + 0.2 * u
        elif role == "beta":
            v = 0.1 * u
        elif role == "rm":
            v = 0.1 * u
        elif role == "rv":
            v = 1.0 + 0.3 * u
        elif role == "nbt":
            out[key] = np.asarray(100, dtype=np.int64)
            continue
        else:
            raise ValueError(role)
        if overrides:
            for sub, f in overrides.items():
                if sub in key:
                    v = v * f
        out[key] = v.astype(np.float32).reshape(shape)
    return out
