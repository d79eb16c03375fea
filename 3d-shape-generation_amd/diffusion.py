"""Diffusion processes with the reference's Python surface (reference diffusion.py):
`PointCloudDiffusion` (:14-358) and `LatentDiffusion` (:361-734).

Host code only sequences the loop: the per-step constants (continuous-time t sequence,
noise/signal rates) are computed on the host CPU with the reference's exact torch ops so
that step indexing is bit-exact, uploaded once per call, and every
per-timestep computation runs in HIP kernels (`_lib`).  The training surface (`training_step`,
`diffusion_loss`, `configure_optimizers`) forwards to the HIP trainers in `training.py`.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .networks import UNetPointNetLarge


class _HParams(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# ------------------------------------------------------------------ host schedule math
def _offset_cosine(t: torch.Tensor, min_signal: float, max_signal: float):
    """diffusion.py:208-223, on CPU tensors."""
    a0 = torch.acos(torch.tensor(max_signal))
    a1 = torch.acos(torch.tensor(min_signal))
    ang = a0 + t * (a1 - a0)
    return torch.sin(ang), torch.cos(ang)


def _linear(t: torch.Tensor, lo: float, hi: float):
    """diffusion.py:189-205 (bug-for-bug: cumprod over the batch axis)."""
    betas = lo + t.clone() * (hi - lo)
    abar = torch.cumprod(1 - betas, dim=0)
    return 1 - abar, abar


class StepTable:
    """Per-step scalars of one sampler run as fp32 device matrices (T, R): R = 1 when every
    shape shares the rates (cosine schedule), R = batch for the linear schedule whose
    batch-axis cumprod (diffusion.py:202) gives each shape its own rates."""

    def __init__(self, t, n, s, a, b, device):
        def f(rows):
            if isinstance(rows, torch.Tensor):                     # already (T, R): the vectorised cosine tables
                return rows.to(torch.float32).reshape(rows.shape[0], -1).contiguous().to(device)
            return torch.stack([torch.as_tensor(x, dtype=torch.float32).reshape(-1) for x in rows]).contiguous().to(device)
        if isinstance(t, torch.Tensor):
            t = t.reshape(t.shape[0], -1)[:, :1]
        else:
            t = [x.reshape(-1)[:1] for x in t]
        self.t, self.n, self.s, self.a, self.b = f(t), f(n), f(s), f(a), f(b)
        self.t = self.t.reshape(-1)
        self.steps = len(t)
        self.width = self.n.shape[1]
        self.stride = 0 if self.width == 1 else 1

    def offset(self, k: int) -> int:
        return 4 * k * self.width


class _DiffusionBase(nn.Module):
    """Schedule + elementwise ops + the three sampler loops, shared by both processes."""

    def _init_schedule(self, noise_schedule: str):
        self.noise_schedule = noise_schedule
        self.linear_min_rate, self.linear_max_rate = 0.0001, 0.02
        self.cosine_min_signal_rate, self.cosine_max_signal_rate = 0.02, 0.95
        self.diffusion_schedule = (self.offset_cosine_diffusion_schedule if noise_schedule == "cosine"
                                   else self.linear_diffusion_schedule)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    # reference diffusion.py:208-223 / 189-205: computed on the host, returned on t's device
    def offset_cosine_diffusion_schedule(self, diffusion_times: torch.Tensor):
        n, s = _offset_cosine(diffusion_times.detach().to("cpu", torch.float32),
                              self.cosine_min_signal_rate, self.cosine_max_signal_rate)
        return n.to(diffusion_times.device), s.to(diffusion_times.device)

    def linear_diffusion_schedule(self, diffusion_times: torch.Tensor):
        n, s = _linear(diffusion_times.detach().to("cpu", torch.float32), self.linear_min_rate, self.linear_max_rate)
        return n.to(diffusion_times.device), s.to(diffusion_times.device)

    # ------------------------------------------------------------------ elementwise ops
    def _rates(self, v: torch.Tensor, batch: int) -> Tuple[torch.Tensor, int]:
        v = v.to(self.device, torch.float32).reshape(-1).contiguous()
        if v.numel() == 1:
            return v, 0
        if v.numel() != batch:
            raise ValueError(f"rates have {v.numel()} entries for a batch of {batch}")
        return v, 1

    # Philox stream layout.  One draw of a (batch, ...) tensor consumes `span` counters (4 normals each); a process
    # that holds samples [lo, lo + b) of a global batch of `total` (set by dist.shard_context) reads the
    # sub-block that starts `lo * per-sample counters` into the draw and advances by the GLOBAL span, so ranks
    # never share counters and (per-sample size divisible by 4) sample i gets the same numbers whatever the
    # number of ranks.
    _shard = None

    def _philox_span(self, numel: int, batch: int) -> Tuple[int, int]:
        per4 = (numel // max(batch, 1) + 3) // 4
        lo, total = self._shard if self._shard is not None else (0, batch)
        return lo * per4, max(total, batch) * per4

    def _randn_like(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x, dtype=torch.float32)
        seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._philox_offset = getattr(self, "_philox_offset", 0)
        batch = x.shape[0] if x.dim() > self._sample_dims else 1
        shard_off, span = self._philox_span(out.numel(), batch)
        _lib.check(_lib.load().pcd_randn(out.data_ptr(), out.numel(), seed, self._philox_offset + shard_off,
                                         _lib.stream_ptr()), "randn")
        self._philox_offset += span
        return out

    def add_noise(self, x_0: torch.Tensor, t: torch.Tensor, noise: Optional[torch.Tensor] = None):
        """diffusion.py:138-152 -> (x_t, noise, noise_rates, signal_rates).  `noise` may be injected."""
        self._require_cuda(x_0)
        x_0 = x_0.to(torch.float32).contiguous()
        if noise is None:
            noise = self._randn_like(x_0)
        noise = noise.to(self.device, torch.float32).contiguous()
        noise_rates, signal_rates = self.diffusion_schedule(t)
        b = x_0.shape[0] if x_0.dim() > self._sample_dims else 1
        n, st = self._rates(noise_rates, b)
        s, _ = self._rates(signal_rates, b)
        x_t = torch.empty_like(x_0)
        _lib.check(_lib.load().pcd_add_noise(x_0.data_ptr(), noise.data_ptr(), n.data_ptr(), s.data_ptr(), st,
                                             x_0.numel(), x_0.numel() // b, x_t.data_ptr(), _lib.stream_ptr()), "add_noise")
        if x_0.dim() == self._sample_dims:   # the reference's rates.view(-1,1,1) broadcast adds a batch axis
            x_t = x_t.unsqueeze(0)
        return x_t, noise, noise_rates, signal_rates

    def remove_noise(self, x_t, predicted_noise, noise_rates, signal_rates):
        """diffusion.py:154-168."""
        self._require_cuda(x_t, predicted_noise)
        x_t = x_t.to(torch.float32).contiguous()
        eps = predicted_noise.to(torch.float32).contiguous()
        b = x_t.shape[0]
        n, st = self._rates(noise_rates, b)
        s, _ = self._rates(signal_rates, b)
        x0 = torch.empty_like(x_t)
        _lib.check(_lib.load().pcd_remove_noise(x_t.data_ptr(), eps.data_ptr(), n.data_ptr(), s.data_ptr(), st,
                                                x_t.numel(), x_t.numel() // b, x0.data_ptr(), _lib.stream_ptr()), "remove_noise")
        return x0

    def _require_cuda(self, *ts):
        if self.device.type != "cuda":
            raise RuntimeError("this framework runs only on an MI355X device: call .to('cuda') first")
        for t in ts:
            if t is not None and t.device != self.device:
                raise RuntimeError(f"tensor on {t.device}, model on {self.device}")
        _lib.load()

    # --------------------------------------------------------------- per-step tables
    def _width(self, batch: int) -> int:
        return 1 if self.noise_schedule == "cosine" else batch

    def _host_schedule(self, t_cpu: torch.Tensor):
        if self.noise_schedule == "cosine":
            return _offset_cosine(t_cpu, self.cosine_min_signal_rate, self.cosine_max_signal_rate)
        return _linear(t_cpu, self.linear_min_rate, self.linear_max_rate)

    def ddim_table(self, num_steps: int, batch: int = 1) -> StepTable:
        """`sample` (diffusion.py:277-286): t_k = 1 - k/T, next_t = t_k - 1/T."""
        w = self._width(batch)
        step = 1.0 / num_steps
        if w == 1 and self.vectorized_tables:
            # all T steps in one set of elementwise ops: the same fp32 operations per element as the loop below
            # (k * step is formed in float64 and rounded once, like the Python scalar in `ones - k * step`), 30 us
            # of host time per step saved; tests/test_oracle_golden.py checks both forms are bit-identical
            t = torch.ones(num_steps) - (torch.arange(num_steps, dtype=torch.float64) * step).to(torch.float32)
            n, s = self._host_schedule(t)
            nn_, sn = self._host_schedule(t - step)
            return StepTable(t, n, s, nn_, sn, self.device)
        ts, ns, ss, n2, s2 = [], [], [], [], []
        for k in range(num_steps):
            t = torch.ones(w) - k * step
            n, s = self._host_schedule(t)
            nn_, sn = self._host_schedule(t - step)
            ts.append(t); ns.append(n); ss.append(s); n2.append(nn_); s2.append(sn)
        return StepTable(ts, ns, ss, n2, s2, self.device)

    def ddpm_table(self, num_steps: int, batch: int = 1) -> StepTable:
        """`sample2` (diffusion.py:241-255): t = i/T for i = T-1..0; a = sqrt(n_prev/n), b = s_prev."""
        w = self._width(batch)
        if w == 1 and self.vectorized_tables:
            i = torch.arange(num_steps - 1, -1, -1, dtype=torch.float32)
            t = torch.ones(num_steps) * i / num_steps
            n, s = self._host_schedule(t)
            npv, sp = self._host_schedule(torch.ones(num_steps) * (i - 1) / num_steps)
            co = torch.sqrt(npv / n)
            co[-1], sp[-1] = 0.0, 0.0                              # i = 0: x_t = x_0, no update
            return StepTable(t, n, s, co, sp, self.device)
        ts, ns, ss, co, s2 = [], [], [], [], []
        for i in reversed(range(num_steps)):
            t = torch.ones(w) * i / num_steps
            n, s = self._host_schedule(t)
            ts.append(t); ns.append(n); ss.append(s)
            if i > 0:
                npv, sp = self._host_schedule(torch.ones(w) * (i - 1) / num_steps)
                co.append(torch.sqrt(npv / n)); s2.append(sp)
            else:
                co.append(torch.zeros(w)); s2.append(torch.zeros(w))
        return StepTable(ts, ns, ss, co, s2, self.device)

    def from_state_table(self, start_t0, num_steps: int) -> StepTable:
        """`sample3` (diffusion.py:323-335): linspace(start_t[0], 0, T); only start_t[0] is used,
        and the schedule sees a 0-d t, so the rates are shared by the batch for both schedules."""
        steps = torch.linspace(torch.as_tensor(start_t0, dtype=torch.float32).cpu().reshape(()),
                               torch.zeros(1)[0], num_steps)
        if self.noise_schedule == "cosine" and self.vectorized_tables:
            n, s = self._host_schedule(steps)
            n2, s2 = torch.zeros(num_steps), torch.zeros(num_steps)
            n2[:-1], s2[:-1] = n[1:], s[1:]
            return StepTable(steps, n, s, n2, s2, self.device)
        ts, ns, ss, n2, s2 = [], [], [], [], []
        for i in range(num_steps):
            n, s = self._host_schedule(steps[i])
            ts.append(steps[i]); ns.append(n); ss.append(s)
            if i < num_steps - 1:
                nn_, sn = self._host_schedule(steps[i + 1])
                n2.append(nn_); s2.append(sn)
            else:
                n2.append(torch.zeros(())); s2.append(torch.zeros(()))
        return StepTable(ts, ns, ss, n2, s2, self.device)

    vectorized_tables = True     # False: per-step host loop (the literal transcription; kept for the equality test)

    # ------------------------------------------------------------------ stepping
    GRAPH_MIN_STEPS = 8
    GRAPH_STEPS = 8            # timesteps captured per HIP graph (one graph launch costs ~10 us of host time: the
    use_graphs = True          # 100 us latent step was launch-bound at one step per graph)

    def _run(self, x, tab: "StepTable", bias_table: torch.Tensor, forward, kind: str, noises=None,
             skip_last_update: bool = False):
        """kind 'ddim' | 'ddpm'.  forward(x, tb_cur, eps_out) enqueues the denoiser for the current step."""
        stp = Stepper(self, x, tab, bias_table, forward, kind, noises)
        T = tab.steps
        last_updates = not (skip_last_update or kind == "ddpm")     # ddpm: x_t = x_0 at i = 0, no update
        n_uniform = T if last_updates else T - 1                       # steps that all look the same
        k = 0
        if self.use_graphs and noises is None and n_uniform - 1 >= self.GRAPH_MIN_STEPS:
            stp.step(0, True)                                          # eager warm-up (loads every kernel)
            per = self.GRAPH_STEPS
            stp.capture(per)
            k = 1
            while k + per <= n_uniform:
                stp.replay()
                k += per
            while k < n_uniform:                                       # remainder: eager, same enqueue
                stp.step(k, True)
                k += 1
        else:
            while k < n_uniform:
                stp.step(k, True)
                k += 1
        if k < T:
            stp.step(k, False)
        if kind == "ddpm":
            self._philox_offset = stp.philox_start + stp.philox_stride * T
        return stp.x0


class Stepper:
    """One timestep = select (device side: copy step k's time bias and rates to fixed buffers, k++) ->
    denoiser forward -> fused update, all on fixed pointers and with the state updated in place, so the
    same enqueue is valid for every k.  Long runs capture it once in a HIP graph and replay it (host
    cost per step: one graph launch instead of ~30 kernel launches)."""

    def __init__(self, owner, x, tab: StepTable, bias_table, forward, kind, noises=None):
        self.lib = _lib.load()
        self.x, self.tab, self.forward, self.kind, self.noises = x, tab, forward, kind, noises
        dev = x.device
        self.T, self.R = tab.steps, tab.width
        self.rates = torch.stack([tab.n, tab.s, tab.a, tab.b]).contiguous()      # (4, T, R)
        self.bias_table = bias_table.contiguous()
        self.tb_elems = self.bias_table.shape[1]
        self.counter = torch.zeros(2, dtype=torch.int32, device=dev)
        self.tb_cur = torch.empty(self.tb_elems, dtype=torch.float32, device=dev)
        self.rates_cur = torch.empty(4 * self.R, dtype=torch.float32, device=dev)
        self.eps = torch.empty_like(x)
        self.x0 = torch.empty_like(x)
        self.z = torch.empty_like(x) if kind == "ddpm" else None
        self.per_shape = x.numel() // x.shape[0]
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        shard_off, span = owner._philox_span(x.numel(), x.shape[0])
        self.philox_start = getattr(owner, "_philox_offset", 0)        # the owner's stream position before this run
        self.philox_base = self.philox_start + shard_off              # this process's sub-block of every draw
        self.philox_stride = span                                     # counters one (global) draw consumes
        self.graph = None

    def step(self, k: int, update: bool = True):
        lib, x, R = self.lib, self.x, self.R
        st = _lib.stream_ptr()
        rp = self.rates_cur.data_ptr()
        _lib.check(lib.pcd_step_select(self.counter.data_ptr(), self.T, self.bias_table.data_ptr(), self.tb_elems,
                                       self.tb_cur.data_ptr(), self.rates.data_ptr(), R, rp, st), "step_select")
        self.forward(x, self.tb_cur, self.eps)
        nxt = x.data_ptr() if update else 0               # in place: every element is read before it is written
        if self.kind == "ddim":
            _lib.check(lib.pcd_ddim_update(x.data_ptr(), self.eps.data_ptr(), rp, rp + 4 * R, rp + 8 * R, rp + 12 * R,
                                           self.tab.stride, x.numel(), self.per_shape, self.x0.data_ptr(), nxt, st),
                       "ddim_update")
            return
        zp = 0
        if update:
            if self.noises is not None:
                self.z.copy_(self.noises[k].to(x.device, torch.float32).reshape(self.z.shape))
            else:
                # on-device noise: the draw and the update are one launch (z is never stored; bitwise pcd_randn_step + pcd_ddpm_update)
                _lib.check(lib.pcd_ddpm_update_philox(x.data_ptr(), self.eps.data_ptr(), rp, rp + 4 * R, rp + 8 * R, rp + 12 * R,
                                                      self.tab.stride, x.numel(), self.per_shape, self.x0.data_ptr(), nxt, self.seed,
                                                      self.philox_base, self.philox_stride, self.counter.data_ptr(), st),
                           "ddpm_update_philox")
                return
            zp = self.z.data_ptr()
        _lib.check(lib.pcd_ddpm_update(x.data_ptr(), self.eps.data_ptr(), zp, rp, rp + 4 * R, rp + 8 * R, rp + 12 * R,
                                       self.tab.stride, x.numel(), self.per_shape, self.x0.data_ptr(), nxt, st),
                   "ddpm_update")

    def capture(self, steps: int = 1):
        """Capture `steps` consecutive generic steps in one graph (must follow at least one eager step: kernels
        loaded, workspaces allocated).  Every step reads its constants through the device-side counter, so the same
        graph is valid wherever it is replayed."""
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(steps):
                self.step(-1, True)

    def replay(self):
        self.graph.replay()


class PointCloudDiffusion(_DiffusionBase):
    """Drop-in for reference diffusion.py:14-358 (sampling surface)."""
    _sample_dims = 2   # one sample is (N, 3)

    def __init__(self, num_points, dim=256, time_dim=256, lr=1e-4, noise_schedule="cosine", backbone="pointnet"):
        """`backbone` is this build's one addition to the reference signature: "pointnet" =
        UNetPointNetLarge, the denoiser diffusion.py:28 wires in; "attention" = UNetAttentionPointExperimental
        (networks.py:597-722), which the reference only reaches by editing the import at diffusion.py:11.  The
        state_dict keys are `model.*` of the chosen class either way."""
        super().__init__()
        self.hparams = _HParams(num_points=num_points, dim=dim, time_dim=time_dim, lr=lr,
                                noise_schedule=noise_schedule)
        if backbone == "pointnet":
            self.model = UNetPointNetLarge(dim, time_dim)
        elif backbone == "attention":
            from .networks import UNetAttentionPointExperimental
            self.model = UNetAttentionPointExperimental(num_points, dim=dim, time_dim=time_dim)
            self.hparams["backbone"] = backbone
        else:
            raise ValueError(f"backbone must be 'pointnet' or 'attention', got {backbone!r}")
        self.backbone = backbone
        self.num_points = num_points
        self.lr = lr
        self._init_schedule(noise_schedule)

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **kwargs):
        """Lightning-free loader for the reference's `.ckpt` layout (test_point_ddpm.py:161)."""
        from .checkpoint import load_lightning_checkpoint
        hp, sd = load_lightning_checkpoint(path, map_location)
        hp.update(kwargs)
        obj = cls(**{k: hp[k] for k in ("num_points", "dim", "time_dim", "lr", "noise_schedule", "backbone") if k in hp})
        obj.load_state_dict(sd, strict=True)
        return obj

    def _forward_fn(self):
        return lambda x, tb_cur, eps: self.model.forward_with_bias(x, tb_cur, 0, out=eps)

    # ------------------------------------------------------------------ training surface (diffusion.py:56-86, 170-186)
    def configure_optimizers(self):
        """AdamW(lr, weight_decay=1e-5) + ReduceLROnPlateau(min, factor 0.5, patience 5) on `val_loss`
        (diffusion.py:60-68); the optimizer object is the HIP trainer, which also owns forward/backward."""
        from .training import PointTrainer, ReduceLROnPlateau
        if self.backbone != "pointnet":
            raise RuntimeError("training is implemented for the reference's wired denoiser (backbone='pointnet') only")
        if getattr(self, "_trainer", None) is None:
            self._trainer = PointTrainer(self.model, lr=self.lr, weight_decay=1e-5)
        return {"optimizer": self._trainer,
                "lr_scheduler": {"scheduler": ReduceLROnPlateau(self._trainer, factor=0.5, patience=5), "monitor": "val_loss"}}

    def diffusion_loss(self, x_0, t, noise=None):
        """diffusion.py:170-186: L1 between the drawn noise and the prediction at x_t.  In train() mode the forward
        uses batch statistics and the parameter gradients are left in the trainer (loss and backward are one pass:
        there is no autograd graph to keep); in eval() mode it is the sampler's folded forward."""
        x_t, noise, _, _ = self.add_noise(x_0, t, noise)
        if self.training:
            tr = self.configure_optimizers()["optimizer"]
            tr.forward(x_t, t.to(self.device, torch.float32))
            return tr.backward(noise)
        pred = self.model(x_t, t.to(self.device, torch.float32))
        lib = _lib.load()
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        scratch = torch.empty_like(pred)
        _lib.check(lib.pcd_l1_loss(pred.data_ptr(), noise.data_ptr(), pred.numel(), 1.0, out.data_ptr(), scratch.data_ptr(),
                                   _lib.stream_ptr()), "l1_loss")
        return out[0] / pred.numel()

    def training_step(self, batch, batch_idx=0):
        """diffusion.py:70-86: t ~ U(0,1) per shape; returns the loss (gradients are ready for `optimizer.step`)."""
        x_0 = batch.to(self.device)
        t = torch.rand(x_0.shape[0], device=self.device)
        return self.diffusion_loss(x_0, t)

    def validation_step(self, batch, batch_idx=0):
        """diffusion.py:88-100 (loss part; the TensorBoard figures of :106-135 are not reproduced)."""
        x_0 = batch.to(self.device)
        t = torch.rand(x_0.shape[0], device=self.device)
        return self.diffusion_loss(x_0, t)

    def _start(self, num_samples, num_points, x_T):
        self.eval()
        self._require_cuda(x_T)
        if x_T is None:
            return self._randn_like(torch.empty(num_samples, num_points, 3, device=self.device))
        if tuple(x_T.shape) != (num_samples, num_points, 3):
            raise ValueError(f"x_T must be {(num_samples, num_points, 3)}, got {tuple(x_T.shape)}")
        return x_T.to(torch.float32).contiguous().clone()

    @torch.no_grad()
    def sample(self, num_samples, num_points, num_steps=1000, x_T=None):
        """DDIM (diffusion.py:261-289).  Returns the last x_0.  `x_T` injects the start noise."""
        x = self._start(num_samples, num_points, x_T)
        tab = self.ddim_table(num_steps, num_samples)
        return self._run(x, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddim")

    @torch.no_grad()
    def sample2(self, num_samples, num_points, num_steps=1000, x_T=None, noises=None):
        """DDPM ancestral sampling (diffusion.py:225-259).  `noises[j]` injects the j-th draw."""
        x = self._start(num_samples, num_points, x_T)
        tab = self.ddpm_table(num_steps, num_samples)
        return self._run(x, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddpm", noises=noises)

    @torch.no_grad()
    def sample3(self, num_samples, num_points, x=None, start_t=None, num_steps=1000):
        """DDIM from a given state/time (diffusion.py:291-337)."""
        self.eval()
        if x is None:
            x = self._start(num_samples, num_points, None)
            start_t = torch.ones(num_samples)
        else:
            x = x.to(self.device, torch.float32).contiguous().clone()
            if start_t is None:
                start_t = torch.ones(num_samples)
        tab = self.from_state_table(start_t.reshape(-1)[0], num_steps)
        return self._run(x, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddim", skip_last_update=True)


class LatentDiffusion(_DiffusionBase):
    """Drop-in for reference diffusion.py:361-734 (sampling surface) over the latents of a frozen VAE."""
    _sample_dims = 1   # one sample is (latent_dim,)

    def __init__(self, vae, latent_dim=256, dim=512, time_dim=256, lr=1e-4, noise_schedule="cosine",
                 is_voxel_based=True):
        super().__init__()
        from .networks import SimpleLatentUNetPointNet
        self.hparams = _HParams(latent_dim=latent_dim, dim=dim, time_dim=time_dim, lr=lr,
                                noise_schedule=noise_schedule, is_voxel_based=is_voxel_based)
        self.vae = vae
        for p in self.vae.parameters():
            p.requires_grad = False
        self.model = SimpleLatentUNetPointNet(latent_dim, dim, time_dim)
        self.lr = lr
        self._init_schedule(noise_schedule)
        self.init_weights()

    def init_weights(self):
        """diffusion.py:392-408, quirk included: the reference's loop skips the child NAMED 'vae' but then
        walks `self.modules()`, which contains the VAE's modules too, so every nn.Linear of the VAE (`fc_mu`,
        `fc_logvar`, `decoder_input`, and VAE3D's `encoder.5`) is re-initialised in place with kaiming-normal
        (fan_out, relu) weights and zero bias when a LatentDiffusion is constructed around it.  Conv3d / BatchNorm3d
        are not in the reference's isinstance lists and stay.  Load VAE weights AFTER constructing LatentDiffusion
        (a LatentDiffusion checkpoint carries the `vae.*` keys, so load_from_checkpoint does)."""
        import math
        sd = dict(self.vae.named_parameters())
        with torch.no_grad():
            for name, w in sd.items():
                if name.endswith(".weight") and w.dim() == 2 and name[:-6] + "bias" in sd:      # an nn.Linear
                    w.normal_(0.0, math.sqrt(2.0 / w.shape[0]))
                    sd[name[:-6] + "bias"].zero_()
        if hasattr(self.vae, "invalidate"):
            self.vae.invalidate()

    @classmethod
    def load_from_checkpoint(cls, path, vae=None, map_location="cpu", **kwargs):
        """`vae=` is required: the reference saves hyper-parameters with ignore=['vae'] (diffusion.py:375)."""
        from .checkpoint import load_lightning_checkpoint
        if vae is None:
            raise TypeError("LatentDiffusion.load_from_checkpoint needs vae=")
        hp, sd = load_lightning_checkpoint(path, map_location)
        hp.update(kwargs)
        keys = ("latent_dim", "dim", "time_dim", "lr", "noise_schedule", "is_voxel_based")
        obj = cls(vae, **{k: hp[k] for k in keys if k in hp})
        obj.load_state_dict(sd, strict=True)
        return obj

    def _forward_fn(self):
        return lambda x, tb_cur, eps: self.model.forward_with_bias(x, tb_cur, 0, out=eps)

    # The DDIM loops (`sample`, `sample3`) of a batch <= 64 run as ONE persistent launch for all their steps
    # (csrc/latent_persist.hip) when this process can have the whole GPU: 256 workgroups, one per CU, each with the whole LDS,
    # must be co-resident.  Otherwise -- larger batches, DDPM (`sample2`), PCD_LATENT_PERSISTENT=0 / use_persistent = False, a CU
    # mask, several ranks of this job on one device -- the per-layer launches run.  A persistent launch that still cannot
    # complete (a foreign process holding CUs: every wait inside is bounded) is not an error: the loop is re-run from the saved
    # start state on the per-layer launches, with one warning, and the module stays on them.
    use_persistent = os.environ.get("PCD_LATENT_PERSISTENT", "1") != "0"

    def _persistent_allowed(self, x) -> bool:
        if not self.use_persistent or x.dim() != 2:
            return False
        # ranks of one job sharing a device (a one-GPU rehearsal of the multi-rank path): two persistent grids cannot be resident together.
        # Said explicitly (PCD_SHARED_GPU=1, or bench.py's PCD_BENCH_SHARE_GPU=1), or inferred when more local ranks exist than devices
        # are visible and no per-rank visibility mask is in play (with a mask every rank sees "one device" that is its own).  Whatever
        # this misses is caught by the fail-soft path below at the price of one abandoned launch (0.2 s) and a warning.
        if os.environ.get("PCD_SHARED_GPU") == "1" or os.environ.get("PCD_BENCH_SHARE_GPU") == "1":
            return False
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", dist.get_world_size()))
            if not masked and local_world > max(torch.cuda.device_count(), 1):
                return False
        return self.model.persist_supported(x.shape[0])

    def _run(self, x, tab, bias_table, forward, kind, noises=None, skip_last_update=False):
        if kind == "ddim" and noises is None and self._persistent_allowed(x):
            # A non-finite start state cannot go through the kernel's exchange (a value IS its own ready flag: NaN / set sign bits
            # mean "not written yet"), so such a call takes the per-layer launches, where GroupNorm keeps the damage inside its row.
            if bool(torch.isfinite(x).all()):
                start = x.clone()                                                        # <= 64 KB: the launch updates x in place
                rates = torch.stack([tab.n, tab.s, tab.a, tab.b]).contiguous()           # (4, T, R)
                counter = torch.zeros(2, dtype=torch.int32, device=x.device)
                x0 = torch.empty_like(x)
                # the last step's update of z is computed and discarded when `skip_last_update` (sample3): x0 is the result
                self.model.ddim_steps_persist(x, x0, bias_table.contiguous(), rates, counter, tab.steps)
                status = self.model.persist_status()
                if status == 0:
                    return x0
                import warnings
                warnings.warn(f"persistent latent kernel abandoned its launch (wait kind {status >> 16}, workgroup {status & 0xffff}): either "
                              "the CUs of this GPU are not all available to this process, or an intermediate value became non-finite (its "
                              "exchange reads NaN / set sign bits as 'not written yet'); re-running on the per-layer launches, which "
                              "propagate non-finite values like the reference, and staying there (LatentDiffusion.use_persistent = False)",
                              RuntimeWarning, stacklevel=3)
                self.use_persistent = False
                x.copy_(start)
        return super()._run(x, tab, bias_table, forward, kind, noises, skip_last_update)

    # ------------------------------------------------------------------ training surface (diffusion.py:410-443, 522-537)
    def configure_optimizers(self, max_epochs: int = 100):
        """AdamW(lr, weight_decay=1e-5) on the denoiser (the VAE is frozen, diffusion.py:377-378) +
        CosineAnnealingLR(T_max=max_epochs, eta_min=1e-6) (diffusion.py:414-419); the optimizer is the HIP trainer."""
        from .training import CosineAnnealingLR, LatentTrainer
        if getattr(self, "_trainer", None) is None:
            self._trainer = LatentTrainer(self.model, lr=self.lr, weight_decay=1e-5)
            self._scheduler = CosineAnnealingLR(self._trainer, T_max=max_epochs, eta_min=1e-6)
        return {"optimizer": self._trainer, "lr_scheduler": self._scheduler}

    def diffusion_loss(self, z_0, t, noise=None, dropout_mask=None):
        """diffusion.py:522-537.  train(): the denoiser runs with Dropout active and the gradients are left in the trainer."""
        z_t, noise, _, _ = self.add_noise(z_0, t, noise)
        if self.training:
            tr = self.configure_optimizers()["optimizer"]
            tr.forward(z_t, t.to(self.device, torch.float32), dropout_mask)
            return tr.backward(noise)
        pred = self.model(z_t, t.to(self.device, torch.float32))
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        scratch = torch.empty_like(pred)
        _lib.check(_lib.load().pcd_l1_loss(pred.data_ptr(), noise.data_ptr(), pred.numel(), 1.0, out.data_ptr(), scratch.data_ptr(),
                                           _lib.stream_ptr()), "l1_loss")
        return out[0] / pred.numel()

    def training_step(self, batch, batch_idx=0):
        """diffusion.py:424-443: z = reparameterize(encode(x)) through the frozen VAE, t ~ U(0,1), L1 loss."""
        x = batch.to(self.device)
        mu, logvar = self.vae.encode(x)
        z = self.vae.reparameterize(mu, logvar)
        t = torch.rand(z.shape[0], device=self.device)
        return self.diffusion_loss(z, t)

    validation_step = training_step        # diffusion.py:445-467: same computation under eval() (figures not reproduced)

    def _start(self, num_samples, z_T):
        self.eval()
        self._require_cuda(z_T)
        if z_T is None:
            return self._randn_like(torch.empty(num_samples, self.hparams.latent_dim, device=self.device))
        return z_T.to(self.device, torch.float32).contiguous().clone()

    def _finish(self, z_0, threshold):
        from .utils import voxel_tensor_to_point_clouds
        x_0 = self.vae.decode(z_0)
        if self.hparams.is_voxel_based:
            return voxel_tensor_to_point_clouds(x_0, threshold=threshold)
        # the reference's sample()/sample3() leave `point_clouds` unbound here (diffusion.py:650-653)
        raise UnboundLocalError("local variable 'point_clouds' referenced before assignment "
                                "(is_voxel_based=False is not supported by the reference's samplers either)")

    @torch.no_grad()
    def sample(self, num_samples, num_steps=1000, threshold=0.4, z_T=None, return_latent=False):
        """DDIM in latent space, VAE decode, voxel -> points (diffusion.py:619-653)."""
        z = self._start(num_samples, z_T)
        tab = self.ddim_table(num_steps, num_samples)
        z0 = self._run(z, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddim")
        pcs = self._finish(z0, threshold)
        return (pcs, z0) if return_latent else pcs

    @torch.no_grad()
    def sample2(self, num_samples, num_steps=1000, threshold=0.4, z_T=None, noises=None, return_latent=False):
        """DDPM in latent space (diffusion.py:575-616)."""
        z = self._start(num_samples, z_T)
        tab = self.ddpm_table(num_steps, num_samples)
        z0 = self._run(z, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddpm", noises=noises)
        pcs = self._finish(z0, threshold)
        return (pcs, z0) if return_latent else pcs

    @torch.no_grad()
    def sample3(self, num_samples, z=None, start_t=None, num_steps=1000, threshold=0.4, return_latent=False):
        """DDIM from a given latent/time (diffusion.py:655-707)."""
        self.eval()
        if z is None:
            z = self._start(num_samples, None)
            start_t = torch.ones(num_samples)
        else:
            z = z.to(self.device, torch.float32).contiguous().clone()
            if start_t is None:
                start_t = torch.ones(num_samples)
        tab = self.from_state_table(start_t.reshape(-1)[0], num_steps)
        z0 = self._run(z, tab, self.model.time_bias(tab.t), self._forward_fn(), "ddim", skip_last_update=True)
        pcs = self._finish(z0, threshold)
        return (pcs, z0) if return_latent else pcs
