"""CPU oracle: a PyTorch-CPU fp32 restatement of the reference's sampler hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg use it, as the checker / the timed CPU baseline.

Every function restates one reference function on plain tensors and a flat
`state_dict` (name -> tensor); no nn.Module, no Lightning.  Citations are into
`/root/reference` (dhillon24/3d-shape-generation @ 2024_10_08).

Parity status: PINNED.  The restatement is checked against golden vectors
captured from the imported reference itself (`oracle/make_golden.py` ->
`tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`) plus the
reference's own `units.py` inputs.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

COS_MIN_SIGNAL = 0.02   # diffusion.py:34
COS_MAX_SIGNAL = 0.95   # diffusion.py:35
LIN_MIN_RATE = 0.0001   # diffusion.py:32
LIN_MAX_RATE = 0.02     # diffusion.py:33


# ------------------------------------------------------------------ schedule
def offset_cosine_schedule(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """diffusion.py:208-223 (latent copy :558-573).  Returns (noise, signal)."""
    a0 = torch.acos(torch.tensor(COS_MAX_SIGNAL))
    a1 = torch.acos(torch.tensor(COS_MIN_SIGNAL))
    ang = a0 + t * (a1 - a0)
    return torch.sin(ang), torch.cos(ang)


def linear_schedule(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """diffusion.py:189-205.  Bug-for-bug: cumprod runs over the *batch* axis."""
    betas = LIN_MIN_RATE + t.clone() * (LIN_MAX_RATE - LIN_MIN_RATE)
    abar = torch.cumprod(1 - betas, dim=0)
    return 1 - abar, abar


def schedule_fn(name: str) -> Callable:
    return offset_cosine_schedule if name == "cosine" else linear_schedule


# ----------------------------------------------------------- time embeddings
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """networks.py:820-838: [sin(t f_j), cos(t f_j)], f_j = exp(-j ln(1e4)/(half-1))."""
    half = dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -step)
    arg = t[:, None] * freqs[None, :]
    emb = torch.cat((torch.sin(arg), torch.cos(arg)), dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def time_mlp(sd: SD, p: str, emb: torch.Tensor) -> torch.Tensor:
    """Linear -> SiLU -> Linear (networks.py:737-741)."""
    h = F.silu(F.linear(emb, sd[p + "time_mlp.0.weight"], sd[p + "time_mlp.0.bias"]))
    return F.linear(h, sd[p + "time_mlp.2.weight"], sd[p + "time_mlp.2.bias"])


# -------------------------------------------------------------- point layers
_BN_TRAIN = False   # set by unet_pointnet_large(train=True): nn.BatchNorm1d in train() mode


def _bn_eval(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    if _BN_TRAIN:   # batch statistics; running estimates updated in place (momentum 0.1, unbiased variance)
        sd[p + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def pointnet_layer(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """networks.py:36-49 on (B, C, N)."""
    for i in (1, 2, 3):
        x = F.conv1d(x, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"])
        x = F.relu(_bn_eval(sd, f"{p}.bn{i}", x))
    return x


def unet_pointnet_large(sd: SD, p: str, x: torch.Tensor, t: torch.Tensor,
                        time_dim: int = 256, taps: Optional[dict] = None, train: bool = False) -> torch.Tensor:
    """networks.py:779-818.  x (B,N,3), t (B,) -> predicted noise (B,N,3).  train=True: the module in train()
    mode (BatchNorm batch statistics; sd's running_* / num_batches_tracked entries are updated in place)."""
    global _BN_TRAIN
    if train != _BN_TRAIN:
        _BN_TRAIN = train
        try:
            return unet_pointnet_large(sd, p, x, t, time_dim, taps, train)
        finally:
            _BN_TRAIN = False
    n = x.shape[1]
    temb = time_mlp(sd, p, timestep_embedding(t, time_dim))
    h = torch.cat([x.transpose(2, 1), temb.unsqueeze(2).expand(-1, -1, n)], dim=1)
    x1 = pointnet_layer(sd, p + "enc1", h)
    x2 = pointnet_layer(sd, p + "enc2", x1)
    x3 = pointnet_layer(sd, p + "enc3", x2)
    x4 = pointnet_layer(sd, p + "enc4", x3)
    g = F.conv1d(x4, sd[p + "global_feat.0.weight"], sd[p + "global_feat.0.bias"])
    g = F.relu(_bn_eval(sd, p + "global_feat.1", g))
    g = F.conv1d(g, sd[p + "global_feat.3.weight"], sd[p + "global_feat.3.bias"])
    g = F.relu(_bn_eval(sd, p + "global_feat.4", g))
    pooled = torch.max(g, 2, keepdim=True)[0]
    g = pooled.repeat(1, 1, n)

    def refine(k, v):
        return F.conv1d(v, sd[f"{p}refine{k}.weight"], sd[f"{p}refine{k}.bias"])

    d4 = pointnet_layer(sd, p + "dec4", torch.cat([g, refine(4, x4)], dim=1))
    d3 = pointnet_layer(sd, p + "dec3", torch.cat([d4, refine(3, x3)], dim=1))
    d2 = pointnet_layer(sd, p + "dec2", torch.cat([d3, refine(2, x2)], dim=1))
    d1 = pointnet_layer(sd, p + "dec1", torch.cat([d2, refine(1, x1)], dim=1))
    o = F.conv1d(d1, sd[p + "output.0.weight"], sd[p + "output.0.bias"])
    o = F.relu(_bn_eval(sd, p + "output.1", o))
    o = F.conv1d(o, sd[p + "output.3.weight"], sd[p + "output.3.bias"])
    if taps is not None:
        taps.update(temb=temb, x1=x1, x2=x2, x3=x3, x4=x4, pooled=pooled.squeeze(2),
                    d4=d4, d3=d3, d2=d2, d1=d1)
    return o.transpose(2, 1)


# ------------------------------------------------------------- set attention
def _layer_norm(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def multihead_self_attention(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(dim, heads) self-attention on (B, N, C), explicit
    bmm/softmax/bmm path (networks.py:61,81): q pre-scaled by 1/sqrt(d)."""
    b, n, c = x.shape
    d = c // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.split(c, dim=-1)

    def split(z):
        return z.reshape(b, n, heads, d).permute(0, 2, 1, 3)

    q = split(q) * (1.0 / math.sqrt(d))
    k = split(k)
    v = split(v)
    w = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def set_attention_block(sd: SD, p: str, x: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """networks.py:70-83 on (B, N, C): x += MHA(LN1 x); x += FF(LN2 x)."""
    x = x + multihead_self_attention(sd, p + "attention", _layer_norm(sd, p + "ln1", x), heads)
    h = _layer_norm(sd, p + "ln2", x)
    h = F.linear(F.relu(F.linear(h, sd[p + "ff.0.weight"], sd[p + "ff.0.bias"])),
                 sd[p + "ff.2.weight"], sd[p + "ff.2.bias"])
    return x + h


def unet_attention(sd: SD, p: str, x: torch.Tensor, t: torch.Tensor,
                   time_dim: int = 256, heads: int = 4, taps: dict = None) -> torch.Tensor:
    """networks.py:652-704 (UNetAttentionPointExperimental.forward).  `taps` (optional dict) receives the three skip tensors
    x1 / x2 / x3 as (B, N, C): the values after att1 + emb2, att2 + emb3 and att3 (networks.py:663-672)."""
    te = time_mlp(sd, p, timestep_embedding(t, time_dim).float())

    def emb(name):
        return F.linear(te, sd[p + name + ".weight"], sd[p + name + ".bias"]).unsqueeze(2)

    def att(name, v):  # v (B,C,N)
        return set_attention_block(sd, p + name + ".", v.transpose(2, 1), heads).transpose(2, 1)

    h = x.transpose(2, 1) + emb("emb1")
    x1 = att("att1", pointnet_layer(sd, p + "enc1", h))
    x1 = x1 + emb("emb2")
    x2 = att("att2", pointnet_layer(sd, p + "enc2", x1))
    x2 = x2 + emb("emb3")
    x3 = att("att3", pointnet_layer(sd, p + "enc3", x2))
    if taps is not None:
        taps.update(x1=x1.transpose(2, 1), x2=x2.transpose(2, 1), x3=x3.transpose(2, 1))
    xb = att("bottleneck", x3)
    xb = att("att_dec3", xb + emb("emb_dec3"))
    h = pointnet_layer(sd, p + "dec3", torch.cat([xb, x3], dim=1))
    h = att("att_dec2", h + emb("emb_dec2"))
    h = pointnet_layer(sd, p + "dec2", torch.cat([h, x2], dim=1))
    h = att("att_dec1", h + emb("emb_dec1"))
    h = pointnet_layer(sd, p + "dec1", torch.cat([h, x1], dim=1))
    h = F.conv1d(h, sd[p + "output.weight"], sd[p + "output.bias"])
    return h.transpose(2, 1)


# ------------------------------------------------------------- latent denoiser
def _lin_gn_relu(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    y = F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"])
    return F.relu(F.group_norm(y, 8, sd[p + ".1.weight"], sd[p + ".1.bias"], 1e-5))


def latent_unet(sd: SD, p: str, z: torch.Tensor, t: torch.Tensor, time_dim: int = 256,
                dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """networks.py:1051-1086.  dropout_mask None = eval mode (Dropout is identity); a (B, 128) keep mask = train()
    mode with that draw of nn.Dropout(0.1) after dec1 (networks.py:1035)."""
    te = time_mlp(sd, p, timestep_embedding(t, time_dim))
    h = torch.cat([z, te], dim=1)
    z1 = _lin_gn_relu(sd, p + "enc1", h)
    z2 = _lin_gn_relu(sd, p + "enc2", z1)
    z3 = _lin_gn_relu(sd, p + "enc3", z2)
    z4 = _lin_gn_relu(sd, p + "enc4", z3)
    g = _lin_gn_relu(sd, p + "global_feat", z4)
    g = F.linear(g, sd[p + "global_feat.3.weight"], sd[p + "global_feat.3.bias"])
    g = F.relu(F.group_norm(g, 8, sd[p + "global_feat.4.weight"], sd[p + "global_feat.4.bias"], 1e-5))

    def refine(k, v):
        return F.linear(v, sd[f"{p}refine{k}.weight"], sd[f"{p}refine{k}.bias"])

    h = _lin_gn_relu(sd, p + "dec4", torch.cat([g, refine(4, z4)], dim=1))
    h = _lin_gn_relu(sd, p + "dec3", torch.cat([h, refine(3, z3)], dim=1))
    h = _lin_gn_relu(sd, p + "dec2", torch.cat([h, refine(2, z2)], dim=1))
    h = _lin_gn_relu(sd, p + "dec1", torch.cat([h, refine(1, z1)], dim=1))
    if dropout_mask is not None:
        h = h * dropout_mask / (1.0 - 0.1)
    h = F.relu(F.linear(h, sd[p + "output.0.weight"], sd[p + "output.0.bias"]))
    return F.linear(h, sd[p + "output.2.weight"], sd[p + "output.2.bias"])


# -------------------------------------------------------------------- VAE3D
def _res3d(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """networks.py:488-504."""
    y = F.conv3d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    y = F.relu(_bn_eval(sd, p + ".bn1", y))
    y = F.conv3d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    y = _bn_eval(sd, p + ".bn2", y)
    if (p + ".downsample.weight") in sd:
        x = F.conv3d(x, sd[p + ".downsample.weight"], sd[p + ".downsample.bias"])
    return F.relu(y + x)


def _run_vae_program(sd: SD, p: str, prog: Sequence, x: torch.Tensor, last_act: str) -> torch.Tensor:
    for i, (idx, op, a) in enumerate(prog):
        key = f"{p}.{idx}"
        last = i == len(prog) - 1
        if op == "res":
            x = _res3d(sd, key, x)
            continue
        if op == "conv":
            x = F.conv3d(x, sd[key + ".weight"], sd[key + ".bias"], stride=a[3], padding=a[4])
        else:
            x = F.conv_transpose3d(x, sd[key + ".weight"], sd[key + ".bias"], stride=a[3], padding=a[4])
        x = (torch.sigmoid(x) if last_act == "sigmoid" else F.relu(x)) if last else F.relu(x)
    return x


def vae_encode(sd: SD, p: str, x: torch.Tensor, enc_prog: Sequence) -> Tuple[torch.Tensor, torch.Tensor]:
    """networks.py:2299-2310 (encoder :2225-2241)."""
    h = _run_vae_program(sd, p + "encoder", enc_prog, x, "relu").flatten(1)
    return (F.linear(h, sd[p + "fc_mu.weight"], sd[p + "fc_mu.bias"]),
            F.linear(h, sd[p + "fc_logvar.weight"], sd[p + "fc_logvar.bias"]))


def vae_decode(sd: SD, p: str, z: torch.Tensor, dec_prog: Sequence) -> torch.Tensor:
    """networks.py:2327-2339 (decoder :2249-2264)."""
    h = F.linear(z, sd[p + "decoder_input.weight"], sd[p + "decoder_input.bias"]).view(-1, 512, 4, 4, 4)
    return _run_vae_program(sd, p + "decoder", dec_prog, h, "sigmoid")


def vae3d_small_encode(sd: SD, p: str, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """VAE3D.encode (networks.py:2044-2057; encoder :1998-2006): four stride-2 Conv3DBlocks, Linear, ReLU."""
    h = x
    for i in range(4):
        h = F.conv3d(h, sd[f"{p}encoder.{i}.conv.weight"], sd[f"{p}encoder.{i}.conv.bias"], stride=2, padding=1)
        h = F.relu(_bn_eval(sd, f"{p}encoder.{i}.bn", h))
    h = F.relu(F.linear(h.flatten(1), sd[p + "encoder.5.weight"], sd[p + "encoder.5.bias"]))
    return (F.linear(h, sd[p + "fc_mu.weight"], sd[p + "fc_mu.bias"]),
            F.linear(h, sd[p + "fc_logvar.weight"], sd[p + "fc_logvar.bias"]))


def vae3d_small_decode(sd: SD, p: str, z: torch.Tensor) -> torch.Tensor:
    """VAE3D.decode (networks.py:2077-2090; decoder :2014-2020)."""
    h = F.linear(z, sd[p + "decoder_input.weight"], sd[p + "decoder_input.bias"]).view(-1, 256, 2, 2, 2)
    for i in range(3):
        h = F.conv_transpose3d(h, sd[f"{p}decoder.{i}.deconv.weight"], sd[f"{p}decoder.{i}.deconv.bias"],
                               stride=2, padding=1, output_padding=1)
        h = F.relu(_bn_eval(sd, f"{p}decoder.{i}.bn", h))
    h = F.conv_transpose3d(h, sd[p + "decoder.3.weight"], sd[p + "decoder.3.bias"], stride=2, padding=1, output_padding=1)
    return torch.sigmoid(h)


def vae_reparameterize(mu: torch.Tensor, logvar: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """networks.py:2312-2325 with the normal draw passed in."""
    return mu + eps * torch.exp(0.5 * logvar)


# ----------------------------------------------------------------- samplers
def _bc(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.view(-1, *([1] * (like.dim() - 1)))


def remove_noise(x_t, eps, n, s):
    """diffusion.py:154-168."""
    return (x_t - _bc(n, x_t) * eps) / _bc(s, x_t)


def add_noise(x0, t, noise, sched=offset_cosine_schedule):
    """diffusion.py:138-152 with the normal draw passed in."""
    n, s = sched(t)
    return _bc(s, x0) * x0 + _bc(n, x0) * noise, n, s


def ddim_sample(model: Callable, x_T: torch.Tensor, num_steps: int,
                sched=offset_cosine_schedule, trace: Optional[list] = None) -> torch.Tensor:
    """`sample` (diffusion.py:261-289 / latent :619-645): returns the LAST x_0."""
    b = x_T.shape[0]
    x = x_T
    step = 1.0 / num_steps
    x0 = x
    for k in range(num_steps):
        t = torch.ones(b) - k * step
        n, s = sched(t)
        eps = model(x, t)
        x0 = remove_noise(x, eps, n, s)
        tn = t - step
        nn_, sn = sched(tn)
        x = _bc(sn, x) * x0 + _bc(nn_, x) * eps
        if trace is not None:
            trace.append((t[0].item(), n[0].item(), s[0].item(), nn_[0].item(), sn[0].item()))
    return x0


def ddpm_sample(model: Callable, x_T: torch.Tensor, num_steps: int, noises: Sequence[torch.Tensor],
                sched=offset_cosine_schedule, trace: Optional[list] = None) -> torch.Tensor:
    """`sample2` (diffusion.py:225-259): ancestral step, noises[j] is the j-th randn_like draw."""
    b = x_T.shape[0]
    x = x_T
    j = 0
    for i in reversed(range(num_steps)):
        t = torch.ones(b) * i / num_steps
        n, s = sched(t)
        eps = model(x, t)
        x0 = remove_noise(x, eps, n, s)
        if i > 0:
            tp = torch.ones(b) * (i - 1) / num_steps
            npv, sp = sched(tp)
            coef = torch.sqrt(npv / n)
            x = _bc(sp, x) * x0 + _bc(coef, x) * _bc(n, x) * noises[j]
            j += 1
            if trace is not None:
                trace.append((t[0].item(), n[0].item(), s[0].item(), npv[0].item(), sp[0].item()))
        else:
            x = x0
            if trace is not None:
                trace.append((t[0].item(), n[0].item(), s[0].item(), float("nan"), float("nan")))
    return x


def ddim_from_state(model: Callable, x: torch.Tensor, start_t: torch.Tensor, num_steps: int,
                    sched=offset_cosine_schedule, trace: Optional[list] = None) -> torch.Tensor:
    """`sample3` (diffusion.py:291-337): linspace(start_t[0], 0, T), scalar t, no update on last."""
    b = x.shape[0]
    steps = torch.linspace(start_t[0], torch.zeros(b)[0], num_steps)
    x0 = x
    for i in range(num_steps):
        t = steps[i]
        n, s = sched(t)
        eps = model(x, t.expand(b))
        x0 = remove_noise(x, eps, n, s)
        if i < num_steps - 1:
            nn_, sn = sched(steps[i + 1])
            x = _bc(sn, x) * x0 + _bc(nn_, x) * eps
            if trace is not None:
                trace.append((t.item(), n.item(), s.item(), nn_.item(), sn.item()))
        elif trace is not None:
            trace.append((t.item(), n.item(), s.item(), float("nan"), float("nan")))
    return x0


# ------------------------------------------------------- voxel <-> points
def voxel_tensor_to_point_clouds(v: torch.Tensor, threshold: float = 0.5) -> List[torch.Tensor]:
    """utils.py:511-539: row-major (z,y,x) scan order, coords 2*i/(dim-1)-1 as [x,y,z]."""
    _, _, d, h, w = v.shape
    out = []
    for i in range(v.shape[0]):
        z, y, x = torch.where(v[i, 0] > threshold)
        if len(z) > 0:
            pts = torch.stack([x, y, z], dim=1).float()
            pts = 2 * pts / torch.tensor([w - 1, h - 1, d - 1]) - 1
        else:
            pts = torch.empty((0, 3))
        out.append(pts)
    return out


def voxelize(points: torch.Tensor, res: int = 32) -> torch.Tensor:
    """utils.py:488-509: ((p+1)*(res-1)/2).long().clamp -> occupancy indexed [x,y,z]."""
    points = points.unsqueeze(0) if points.dim() == 2 else points
    idx = ((points + 1) * (res - 1) / 2).long().clamp(0, res - 1)
    vox = torch.zeros(points.size(0), res, res, res)
    for i in range(points.size(0)):
        vox[i, idx[i, :, 0], idx[i, :, 1], idx[i, :, 2]] = 1
    return vox


# ------------------------------------------------------------------ metrics
def normalize_to_cube(p: torch.Tensor) -> torch.Tensor:
    """metrics.py:7-21."""
    c = (p.max(dim=1, keepdim=True)[0] + p.min(dim=1, keepdim=True)[0]) / 2
    p = p - c
    scale = p.abs().max(dim=1, keepdim=True)[0].max(dim=2, keepdim=True)[0]
    return p / scale


def _batched(x):
    return x.unsqueeze(0) if x.dim() == 2 else x


def chamfer_distance(x, y, scaling_factor=1e3):
    """metrics.py:23-47: unsquared L2 via torch.cdist, one scalar for the batch."""
    x, y = normalize_to_cube(_batched(x)), normalize_to_cube(_batched(y))
    d = torch.cdist(x, y)
    return (d.min(dim=2)[0].mean() + d.min(dim=1)[0].mean()) * scaling_factor


def chamfer_distance_exact(x, y, scaling_factor=1e3):
    """Same definition evaluated by direct differences in float64 (no matmul
    cancellation): the accuracy yardstick for the HIP kernel."""
    x = normalize_to_cube(_batched(x)).double()
    y = normalize_to_cube(_batched(y)).double()
    d = (x[:, :, None, :] - y[:, None, :, :]).pow(2).sum(-1).sqrt()
    return (d.min(dim=2)[0].mean() + d.min(dim=1)[0].mean()) * scaling_factor


def earth_mover_distance_cpu(x, y, scaling_factor=1):
    """metrics.py:49-92.  Bug-for-bug: divides by shape[1] of an (N,3) cloud, i.e. 3."""
    from scipy.optimize import linear_sum_assignment
    x, y = normalize_to_cube(_batched(x)), normalize_to_cube(_batched(y))
    vals = []
    for xp, yp in zip(x, y):
        a, b = xp.numpy(), yp.numpy()
        dist = np.linalg.norm(a[:, None] - b[None, :], axis=-1)
        r, c = linear_sum_assignment(dist)
        vals.append(dist[r, c].sum() / max(xp.shape[1], yp.shape[1]))
    return torch.tensor(vals).mean() * scaling_factor


def earth_mover_distance_sinkhorn(x, y, epsilon=1e-2, thresh=1e-5, max_iter=100, scaling_factor=1):
    """metrics.py:94-158 (`earth_mover_distance_gpu`)."""
    x, y = normalize_to_cube(_batched(x)), normalize_to_cube(_batched(y))
    b, n, _ = x.shape
    m = y.shape[1]
    C = torch.cdist(x, y, p=2)
    C = C / C.max()
    lam = 1 / epsilon
    alpha = torch.zeros(b, n, 1)
    beta = torch.zeros(b, m, 1)
    mu = torch.ones(b, n, 1) / n
    nu = torch.ones(b, m, 1) / m
    for _ in range(max_iter):
        a_prev, b_prev = alpha, beta
        alpha = epsilon * (torch.log(mu + 1e-10)
                           - torch.logsumexp(-lam * C + beta.transpose(1, 2), dim=2, keepdim=True))
        beta = epsilon * (torch.log(nu + 1e-10)
                          - torch.logsumexp(-lam * C.transpose(1, 2) + alpha.transpose(1, 2), dim=2, keepdim=True))
        if (alpha - a_prev).abs().max() < thresh and (beta - b_prev).abs().max() < thresh:
            break
    P = torch.exp(-lam * C + alpha + beta.transpose(1, 2))
    return torch.sum(P * C, dim=(1, 2)).mean() * scaling_factor


def compute_metrics(gen, ref, use_approximate_gpu_emd=False):
    """metrics.py:160-183 -> (chamfer x1e3, emd, voxel BCE)."""
    cd = chamfer_distance(gen, ref)
    emd = (earth_mover_distance_sinkhorn if use_approximate_gpu_emd else earth_mover_distance_cpu)(gen, ref)
    rec = F.binary_cross_entropy(voxelize(gen), voxelize(ref))
    return cd, emd, rec


# ------------------------------------------------------------------ training step (diffusion.py:56-86,170-186)
def point_training_step(sd: SD, p: str, x_t: torch.Tensor, t: torch.Tensor, noise: torch.Tensor):
    """diffusion_loss after add_noise: loss = F.l1_loss(noise, model(x_t, t)) with the model in train() mode, and
    its gradients by autograd.  Returns (loss, {key: grad}); sd's BatchNorm running statistics are updated in
    place like the reference's buffers.  Parameters = every floating entry that is not a running statistic."""
    work = dict(sd)
    leaves = {}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
    with torch.enable_grad():
        pred = unet_pointnet_large(work, p, x_t, t, train=True)
        loss = F.l1_loss(noise, pred)
        loss.backward()
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            sd[k] = work[k]
    return loss.detach(), {k: v.grad for k, v in leaves.items()}


def adamw_step(params: SD, grads: SD, state: dict, lr: float = 1e-4, weight_decay: float = 1e-5,
               betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.AdamW as configured at diffusion.py:60, restated: decoupled decay, bias-corrected moments."""
    state["step"] = state.get("step", 0) + 1
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** state["step"], 1 - b2 ** state["step"]
    for k, w in params.items():
        if k not in grads:
            continue
        g = grads[k]
        m = state.setdefault("m", {}).setdefault(k, torch.zeros_like(w))
        v = state.setdefault("v", {}).setdefault(k, torch.zeros_like(w))
        w.mul_(1 - lr * weight_decay)
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        w.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)


def latent_training_step(sd: SD, p: str, z_t: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, dropout_mask: torch.Tensor):
    """LatentDiffusion.diffusion_loss (diffusion.py:522-537) with the denoiser in train() mode and its gradients by
    autograd; only the entries under prefix p (the denoiser; the VAE is frozen, diffusion.py:377-378) get gradients."""
    work = dict(sd)
    leaves = {}
    for k, v in sd.items():
        if k.startswith(p) and v.is_floating_point():
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
    with torch.enable_grad():
        pred = latent_unet(work, p, z_t, t, dropout_mask=dropout_mask)
        loss = F.l1_loss(noise, pred)
        loss.backward()
    return loss.detach(), pred.detach(), {k: v.grad for k, v in leaves.items()}


# ------------------------------------------------------------------ VAE training step (networks.py:2341-2396)
def vae_training_step(sd: SD, p: str, x: torch.Tensor, eps: torch.Tensor, kl_weight: float, enc_prog, dec_prog):
    """VAE3DLarge.calculate_loss(mode='train') with the module in train() mode (BatchNorm3d batch statistics, running
    estimates updated in sd) for a given reparameterisation draw eps and KL weight, and its gradients by autograd.
    Returns (loss, recon_loss, kl_div, recon, mu, logvar, {key: grad})."""
    global _BN_TRAIN
    work = dict(sd)
    leaves = {}
    for k, v in sd.items():
        if k.startswith(p) and v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
    _BN_TRAIN = True
    try:
        with torch.enable_grad():
            mu, logvar = vae_encode(work, p, x, enc_prog)
            z = mu + eps * torch.exp(0.5 * logvar)
            recon = vae_decode(work, p, z, dec_prog)
            recon_loss = F.binary_cross_entropy(recon, x, reduction="mean")
            kl = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
            loss = recon_loss + kl_weight * kl
            loss.backward()
    finally:
        _BN_TRAIN = False
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            sd[k] = work[k]
    return (loss.detach(), recon_loss.detach(), kl.detach(), recon.detach(), mu.detach(), logvar.detach(),
            {k: v.grad for k, v in leaves.items()})


def vae_kl_weight(epoch: int, max_epochs: int, kl_warmup_epochs: int = 10, kl_warmup_max_beta: float = 0.1,
                  kl_annealing_epochs: int = 100) -> float:
    """VAE3DLarge.get_kl_weight (networks.py:2355-2370), including its hard-coded `current_epoch < 10`."""
    annealing = min(max_epochs, kl_annealing_epochs)
    if epoch < 10:
        return (epoch + 1) / kl_warmup_epochs * kl_warmup_max_beta
    return min(kl_warmup_max_beta + (epoch - kl_warmup_epochs + 1) / (annealing - kl_warmup_epochs) * (1.0 - kl_warmup_max_beta), 1.0)
