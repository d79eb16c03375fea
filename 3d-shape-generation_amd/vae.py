"""VAE3DLarge (reference networks.py:2208-2489): voxel VAE with the reference's
encode / reparameterize / decode / forward / sample API on HIP implicit-GEMM convolutions.

Activations are NDHWC fp16 on the device; eval-mode BatchNorm3d is folded into the conv
weights at pack time; ConvTranspose3d(k4,s2,p1) runs as 8 output-parity classes of 2x2x2 taps.
The training surface (`calculate_loss`, `training_step`, `configure_optimizers`) forwards to `training_vae.VAETrainer`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib, packing, specs
from .networks import _HipModule, _dev16, _dev32
from .utils import voxel_tensor_to_point_clouds


def _launch_convs(descs: List["_lib.Conv3dDesc"], device, what: str) -> None:
    """One pcd_conv3d_f16_multi launch for `descs` (problems that differ only in taps / weights / output
    parity), with the split-K scratch the library asks for."""
    lib = _lib.load()
    arr = (_lib.Conv3dDesc * len(descs))(*descs)
    need = int(lib.pcd_conv3d_workspace_bytes(arr, len(descs)))
    ws = torch.empty(need, dtype=torch.uint8, device=device) if need else None
    _lib.check(lib.pcd_conv3d_f16_multi(arr, len(descs), _lib.ptr(ws), need, _lib.stream_ptr()), what)


def _taps_regular(k: int, pad: int) -> np.ndarray:
    t = []
    for kz in range(k):
        for ky in range(k):
            for kx in range(k):
                t.append(((kz - pad) & 0xff) | (((ky - pad) & 0xff) << 8) | (((kx - pad) & 0xff) << 16))
    return np.asarray(t, np.int32)


def _pack_conv(w: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray, int]:
    """(cout, cin, k,k,k) -> [cout][tap*cin] zero-padded to a multiple of 64."""
    cout, cin = w.shape[:2]
    wk = np.transpose(w, (0, 2, 3, 4, 1)).reshape(cout, -1)
    kpad = (wk.shape[1] + 63) // 64 * 64
    out = np.zeros((cout, kpad))
    out[:, :wk.shape[1]] = wk
    return out, b, kpad


# per dimension: output parity -> [(kernel index, input offset)]   (o = 2 i - 1 + k)
_CT_TAPS = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}


def _pack_convT_class(w: np.ndarray, pz: int, py: int, px: int):
    """ConvTranspose3d weight (cin, cout, 4,4,4) -> class matrix [cout][8*cin] + tap table."""
    cin, cout = w.shape[:2]
    cols, taps = [], []
    for kz, dz in _CT_TAPS[pz]:
        for ky, dy in _CT_TAPS[py]:
            for kx, dx in _CT_TAPS[px]:
                cols.append(w[:, :, kz, ky, kx].T)          # [cout][cin]
                taps.append((dz & 0xff) | ((dy & 0xff) << 8) | ((dx & 0xff) << 16))
    return np.ascontiguousarray(np.concatenate(cols, axis=1)), np.asarray(taps, np.int32)


def _shard_span(module, draw: torch.Tensor) -> Tuple[int, int]:
    """(offset of this process's rows inside one global draw, Philox counters the global draw consumes): the
    same layout as diffusion._philox_span; `module._shard = (lo, total)` is set by dist.shard_context."""
    b = max(draw.shape[0], 1)
    per4 = (draw.numel() // b + 3) // 4
    lo, total = getattr(module, "_shard", None) or (0, b)
    return lo * per4, max(total, b) * per4


class VAE3DLarge(_HipModule):
    """Drop-in for reference networks.py:2208-2489 (inference surface)."""

    def __init__(self, input_shape=(32, 32, 32), latent_dim=256, lr=1e-4, kl_warmup_epochs=10,
                 kl_warmup_max_beta=0.1, kl_annealing_epochs=100):
        super().__init__()
        from .diffusion import _HParams
        self.hparams = _HParams(input_shape=input_shape, latent_dim=latent_dim, lr=lr,
                                kl_warmup_epochs=kl_warmup_epochs, kl_warmup_max_beta=kl_warmup_max_beta,
                                kl_annealing_epochs=kl_annealing_epochs)
        if tuple(input_shape) != (32, 32, 32):
            raise ValueError("VAE3DLarge's encoder reduces 32^3 to 1^3; other input shapes fail in the reference too")
        self.latent_dim = latent_dim
        self._build_from_spec(specs.vae3d_large_spec(latent_dim))
        self._handle = None
        # the residual blocks' 1x1x1 projection shortcuts inside their conv2 launches (csrc/conv3d.hip, second source);
        # PCD_VAE_FUSE_SHORTCUT=0 / set_fuse_shortcut(False): a pointwise launch + a residual read per block, as before round 4
        self.fuse_shortcut = os.environ.get("PCD_VAE_FUSE_SHORTCUT", "1") != "0"
        # reference init_weights (networks.py:2281-2283): xavier-normal(gain 0.01) for the latent heads
        with torch.no_grad():
            for name in ("fc_mu", "fc_logvar"):
                w = self._modules[name].weight
                w.normal_(0.0, 0.01 * (2.0 / (w.shape[0] + w.shape[1])) ** 0.5)

    def set_fuse_shortcut(self, on: bool) -> "VAE3DLarge":
        if bool(on) != self.fuse_shortcut:
            self.invalidate()
            self.fuse_shortcut = bool(on)
        return self

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **kwargs):
        """Lightning-free loader for the reference's `.ckpt` layout (test_point_ldm.py:156, train_point_ldm.py:191)."""
        from .checkpoint import load_lightning_checkpoint
        hp, sd = load_lightning_checkpoint(path, map_location)
        hp.update(kwargs)
        keys = ("input_shape", "latent_dim", "lr", "kl_warmup_epochs", "kl_warmup_max_beta", "kl_annealing_epochs")
        obj = cls(**{k: hp[k] for k in keys if k in hp})
        obj.load_state_dict(sd, strict=True)
        return obj

    # ---------------------------------------------------------------- training surface (networks.py:2286-2416)
    current_epoch = 0          # set by training.fit, like Lightning does
    _max_epochs = 100

    def get_kl_weight(self) -> float:
        """networks.py:2355-2370: linear warm-up to kl_warmup_max_beta, then annealing to 1 (incl. the hard-coded `< 10`)."""
        hp = self.hparams
        annealing_epochs = min(self._max_epochs, hp.kl_annealing_epochs)
        if self.current_epoch < 10:
            return (self.current_epoch + 1) / hp.kl_warmup_epochs * hp.kl_warmup_max_beta
        return min(hp.kl_warmup_max_beta + (self.current_epoch - hp.kl_warmup_epochs + 1) /
                   (annealing_epochs - hp.kl_warmup_epochs) * (1.0 - hp.kl_warmup_max_beta), 1.0)

    def configure_optimizers(self):
        """torch.optim.Adam(lr) + ReduceLROnPlateau(min, 0.5, patience 5) on `val_loss` (networks.py:2286-2297);
        the optimizer object is the HIP trainer (`training_vae.VAETrainer`), which also owns forward / backward."""
        from .training import ReduceLROnPlateau
        from .training_vae import VAETrainer
        if getattr(self, "_trainer", None) is None:
            self._trainer = VAETrainer(self, lr=self.hparams.lr)
        return {"optimizer": self._trainer,
                "lr_scheduler": {"scheduler": ReduceLROnPlateau(self._trainer, factor=0.5, patience=5), "monitor": "val_loss"}}

    def calculate_loss(self, batch, mode):
        """networks.py:2372-2402 -> (loss, reconstruction).  mode 'train' in train(): batch statistics, gradients left in
        the trainer, KL weight from the schedule; otherwise the sampler's eval path and KL weight 1."""
        x = batch.to(self.device, torch.float32)
        lib = _lib.load()
        if mode == "train" and self.training:
            tr = self.configure_optimizers()["optimizer"]
            tr.forward(x)
            loss, _, _ = tr.backward(self.get_kl_weight())
            return loss, tr.recon
        recon, mu, logvar = self(x)
        out = torch.empty(2, dtype=torch.float32, device=self.device)
        _lib.check(lib.pcd_binary_bce_mean(recon.data_ptr(), x.contiguous().data_ptr(), recon.numel(), out.data_ptr(), _lib.stream_ptr()), "bce")
        zero = torch.zeros_like(mu)
        scratch = torch.empty_like(mu)
        _lib.check(lib.pcd_vae_latent_backward(mu.data_ptr(), logvar.data_ptr(), zero.data_ptr(), zero.data_ptr(), mu.numel(), 0.0,
                                               scratch.data_ptr(), scratch.data_ptr(), out[1:].data_ptr(), _lib.stream_ptr()), "kl")
        kl = -0.5 * out[1] / mu.numel()
        w = self.get_kl_weight() if mode == "train" else 1.0
        return out[0] + w * kl, recon

    def training_step(self, batch, batch_idx=0):
        return self.calculate_loss(batch, mode="train")[0]

    def validation_step(self, batch, batch_idx=0):
        return self.calculate_loss(batch, mode="val")[0]       # networks.py:2416-2444 minus the TensorBoard figures

    # ---------------------------------------------------------------- packing
    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        dev = self._need_cuda()
        _lib.require_gpu()
        sd = self.state_dict()
        g = lambda k: sd[k].detach().to("cpu", torch.float64).numpy()
        pk: Dict[str, object] = {"zero": torch.zeros(64, dtype=torch.float16, device=dev)}
        pk["taps3"] = torch.from_numpy(_taps_regular(3, 1)).to(dev)
        pk["taps4s2"] = torch.from_numpy(_taps_regular(4, 1)).to(dev)
        pk["taps4p0"] = torch.from_numpy(_taps_regular(4, 0)).to(dev)
        pk["taps1"] = torch.from_numpy(_taps_regular(1, 0)).to(dev)

        def fold(key, bn):
            w, b = g(key + ".weight"), g(key + ".bias")
            if bn is not None:
                scale = g(bn + ".weight") / np.sqrt(g(bn + ".running_var") + packing.BN_EPS)
                w = w * scale[:, None, None, None, None]
                b = (b - g(bn + ".running_mean")) * scale + g(bn + ".bias")
            return w, b

        def conv(key, bn=None):
            w, b = fold(key, bn)
            wk, b, kpad = _pack_conv(w, b)
            return {"w": _dev16(wk, dev), "b": _dev32(b, dev), "kpad": kpad, "cin": w.shape[1], "cout": w.shape[0],
                    "k": w.shape[2]}

        def res(key):
            d = {"c1": conv(key + ".conv1", key + ".bn1"), "c2": conv(key + ".conv2", key + ".bn2")}
            if (key + ".downsample.weight") in sd:
                d["ds"] = conv(key + ".downsample")
                if self.fuse_shortcut:
                    # relu(bn2(conv2 h) + downsample(x)) = relu([gather(h) | x] . [W2 | Wds]^T + b2 + bds): the shortcut's weights as K
                    # columns behind conv2's 27 taps, x as the launch's second source (pcd_conv3d_desc_t.in2) -- no shortcut tensor
                    w2, b2 = fold(key + ".conv2", key + ".bn2")
                    wd, bd = fold(key + ".downsample", None)
                    wk = np.concatenate([np.transpose(w2, (0, 2, 3, 4, 1)).reshape(w2.shape[0], -1), wd.reshape(wd.shape[0], -1)], axis=1)
                    kpad = (wk.shape[1] + 63) // 64 * 64
                    wp = np.zeros((wk.shape[0], kpad))
                    wp[:, :wk.shape[1]] = wk
                    d["c2"] = {"w": _dev16(wp, dev), "b": _dev32(b2 + bd, dev), "kpad": kpad, "cin": w2.shape[1], "cout": w2.shape[0], "k": 3}
                    d["fused"] = True
            return d

        def convT(key):
            w, b = g(key + ".weight"), g(key + ".bias")
            classes = []
            for pz in (0, 1):
                for py in (0, 1):
                    for px in (0, 1):
                        wc, taps = _pack_convT_class(w, pz, py, px)
                        classes.append({"w": _dev16(wc, dev), "taps": torch.from_numpy(taps).to(dev), "p": (pz, py, px)})
            return {"classes": classes, "b": _dev32(b, dev), "cin": w.shape[0], "cout": w.shape[1]}

        w0, b0 = g("encoder.0.weight"), g("encoder.0.bias")
        pk["enc0_w"], pk["enc0_b"] = _dev32(w0.reshape(w0.shape[0], 27), dev), _dev32(b0, dev)
        for idx, op, a in specs.VAE_ENC[1:]:
            pk[f"enc{idx}"] = res(f"encoder.{idx}") if op == "res" else conv(f"encoder.{idx}")
        wl = np.concatenate([g("fc_mu.weight"), g("fc_logvar.weight")], axis=0)
        bl = np.concatenate([g("fc_mu.bias"), g("fc_logvar.bias")], axis=0)
        pk["fc_w"], pk["fc_b"] = _dev16(wl, dev), _dev32(bl, dev)
        # decoder_input: output (c, z, y, x) -> NDHWC order (z, y, x, c)
        wd, bd = g("decoder_input.weight"), g("decoder_input.bias")
        perm = np.arange(512 * 64).reshape(512, 64).T.reshape(-1)
        pk["din_w"], pk["din_b"] = _dev16(wd[perm], dev), _dev32(bd[perm], dev)
        for idx, op, a in specs.VAE_DEC[:-1]:
            key = f"decoder.{idx}"
            pk[f"dec{idx}"] = res(key) if op == "res" else (convT(key) if op == "convT" else conv(key))
        wl_, bl_ = g("decoder.12.weight"), g("decoder.12.bias")
        pk["last_w"] = _dev32(np.transpose(wl_[0], (1, 2, 3, 0)).reshape(27, -1), dev)
        pk["last_b"] = float(bl_[0])
        # the C descriptor of pcd_vae_create: the encode / decode programs run behind one handle
        def cdesc(dst, L):
            dst.w, dst.b, dst.kpad, dst.cin, dst.cout, dst.k = L["w"].data_ptr(), L["b"].data_ptr(), L["kpad"], L["cin"], L["cout"], L["k"]

        def rdesc(dst, R):
            cdesc(dst.c1, R["c1"]); cdesc(dst.c2, R["c2"])
            dst.has_ds = 1 if "ds" in R else 0
            dst.fused_ds = 1 if R.get("fused") else 0
            if "ds" in R:
                cdesc(dst.ds, R["ds"])

        def tdesc(dst, T):
            for cls in T["classes"]:
                pz, py, px = cls["p"]
                k = 4 * pz + 2 * py + px
                dst.w[k], dst.taps[k] = cls["w"].data_ptr(), cls["taps"].data_ptr()
            dst.b, dst.cin, dst.cout = T["b"].data_ptr(), T["cin"], T["cout"]

        d = _lib.VaeDesc()
        d.latent_dim = self.latent_dim
        d.enc0_w, d.enc0_b = pk["enc0_w"].data_ptr(), pk["enc0_b"].data_ptr()
        for i, idx in enumerate((2, 5, 8, 11)):
            rdesc(d.enc_res[i], pk[f"enc{idx}"])
            rdesc(d.dec_res[i], pk[f"dec{idx}"])
        for i, idx in enumerate((3, 6, 9)):
            cdesc(d.enc_down[i], pk[f"enc{idx}"])
        for i, idx in enumerate((0, 3, 6)):
            tdesc(d.dec_up[i], pk[f"dec{idx}"])
        cdesc(d.enc_last, pk["enc12"])
        cdesc(d.dec_conv9, pk["dec9"])
        d.fc_w, d.fc_b, d.din_w, d.din_b = (pk[k].data_ptr() for k in ("fc_w", "fc_b", "din_w", "din_b"))
        d.last_w, d.last_b = pk["last_w"].data_ptr(), pk["last_b"]
        d.taps3, d.taps4s2, d.taps4p0, d.taps1 = (pk[k].data_ptr() for k in ("taps3", "taps4s2", "taps4p0", "taps1"))
        d.zero_page = pk["zero"].data_ptr()
        handle = C.c_void_p()
        _lib.check(_lib.load().pcd_vae_create(C.byref(d), C.byref(handle)), "vae_create")
        self._handle = handle
        self._packed = pk
        return pk

    def _release(self):
        if getattr(self, "_handle", None):
            _lib.load().pcd_vae_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---------------------------------------------------------------- reference API
    def encode(self, x: torch.Tensor):
        """networks.py:2299-2310: (B,1,32,32,32) -> (mu, logvar), each (B, latent_dim).  One pcd_vae_encode call."""
        self._need_cuda(x)
        self._ensure_packed()
        lib = _lib.load()
        b = x.shape[0]
        if tuple(x.shape[1:]) != (1, 32, 32, 32):
            raise ValueError(f"x must be (B, 1, 32, 32, 32), got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        out = torch.empty(b, 2 * self.latent_dim, dtype=torch.float32, device=x.device)
        ws = self._workspace((b,), lib.pcd_vae_workspace_bytes(b))
        _lib.check(lib.pcd_vae_encode(self._handle, x.data_ptr(), b, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr()), "vae_encode")
        return out[:, :self.latent_dim].contiguous(), out[:, self.latent_dim:].contiguous()

    def reparameterize(self, mu, logvar, eps=None):
        """networks.py:2312-2325; `eps` injects the normal draw (otherwise on-device Philox)."""
        self._need_cuda(mu, logvar)
        lib = _lib.load()
        mu, logvar = mu.to(torch.float32).contiguous(), logvar.to(torch.float32).contiguous()
        if eps is None:
            eps = torch.empty_like(mu)
            self._philox = getattr(self, "_philox", 0)
            off, span = _shard_span(self, eps)
            _lib.check(lib.pcd_randn(eps.data_ptr(), eps.numel(), int(torch.initial_seed()) & (2 ** 64 - 1),
                                     (1 << 40) + self._philox + off, _lib.stream_ptr()), "randn")
            self._philox += span
        eps = eps.to(mu.device, torch.float32).contiguous()
        z = torch.empty_like(mu)
        _lib.check(lib.pcd_reparameterize(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), z.data_ptr(), z.numel(),
                                          _lib.stream_ptr()), "reparameterize")
        return z

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """networks.py:2327-2339: (B, latent_dim) -> occupancy probabilities (B,1,32,32,32) fp32.  One pcd_vae_decode call."""
        self._need_cuda(z)
        self._ensure_packed()
        lib = _lib.load()
        b = z.shape[0]
        z = z.to(torch.float32).contiguous()
        out = torch.empty(b, 1, 32, 32, 32, dtype=torch.float32, device=z.device)
        ws = self._workspace((b,), lib.pcd_vae_workspace_bytes(b))
        _lib.check(lib.pcd_vae_decode(self._handle, z.data_ptr(), b, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr()), "vae_decode")
        return out

    def forward(self, x, eps=None):
        """networks.py:2341-2353 -> (reconstruction, mu, logvar)."""
        mu, logvar = self.encode(x)
        return self.decode(self.reparameterize(mu, logvar, eps)), mu, logvar

    @torch.no_grad()
    def sample(self, num_samples, threshold=0.4, z=None):
        """networks.py:2446-2462 -> python list of (n_i, 3) point clouds."""
        self.eval()
        if z is None:
            z = torch.empty(num_samples, self.latent_dim, device=self.device)
            self._philox_sample = getattr(self, "_philox_sample", 0)      # advances like torch.randn's generator does
            off, span = _shard_span(self, z)
            _lib.check(_lib.load().pcd_randn(z.data_ptr(), z.numel(), int(torch.initial_seed()) & (2 ** 64 - 1),
                                             (1 << 41) + self._philox_sample + off, _lib.stream_ptr()), "randn")
            self._philox_sample += span
        return voxel_tensor_to_point_clouds(self.decode(z), threshold)


# per dimension for ConvTranspose3d(k3, s2, p1, output_padding 1): output parity -> [(kernel index, input offset)]
_CT3_TAPS = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}


def _pack_convT3_class(w: np.ndarray, pz: int, py: int, px: int):
    """ConvTranspose3d weight (cin, cout, 3,3,3) -> class matrix [cout][ntaps*cin] (K padded to 64) + taps."""
    cols, taps = [], []
    for kz, dz in _CT3_TAPS[pz]:
        for ky, dy in _CT3_TAPS[py]:
            for kx, dx in _CT3_TAPS[px]:
                cols.append(w[:, :, kz, ky, kx].T)
                taps.append((dz & 0xff) | ((dy & 0xff) << 8) | ((dx & 0xff) << 16))
    wc = np.concatenate(cols, axis=1)
    kpad = (wc.shape[1] + 63) // 64 * 64
    out = np.zeros((wc.shape[0], kpad))
    out[:, :wc.shape[1]] = wc
    return out, np.asarray(taps, np.int32), kpad


class VAE3D(_HipModule):
    """Drop-in for the small voxel VAE, reference networks.py:1984-2206 (inference surface): stride-2
    Conv3DBlocks down to 256 x 2^3, Linear(2048, 512), mirrored Deconv3DBlocks, on the same HIP
    implicit-GEMM kernel as VAE3DLarge."""

    def __init__(self, input_shape=(32, 32, 32), latent_dim=256, beta=1e-1):
        super().__init__()
        from .diffusion import _HParams
        self.hparams = _HParams(input_shape=input_shape, latent_dim=latent_dim, beta=beta)
        if tuple(input_shape) != (32, 32, 32):
            raise ValueError("VAE3D flattens 256 x 2 x 2 x 2 features: only 32^3 grids fit its Linear(2048, 512)")
        self.latent_dim = latent_dim
        self._build_from_spec(specs.vae3d_small_spec(latent_dim))

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **kwargs):
        from .checkpoint import load_lightning_checkpoint
        hp, sd = load_lightning_checkpoint(path, map_location)
        hp.update(kwargs)
        obj = cls(**{k: hp[k] for k in ("input_shape", "latent_dim", "beta") if k in hp})
        obj.load_state_dict(sd, strict=True)
        return obj

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        dev = self._need_cuda()
        _lib.require_gpu()
        sd = self.state_dict()
        g = lambda k: sd[k].detach().to("cpu", torch.float64).numpy()
        pk = {"zero": torch.zeros(64, dtype=torch.float16, device=dev),
              "taps3": torch.from_numpy(_taps_regular(3, 1)).to(dev)}

        def bn_scale(bn):
            scale = g(bn + ".weight") / np.sqrt(g(bn + ".running_var") + packing.BN_EPS)
            return scale, g(bn + ".bias") - g(bn + ".running_mean") * scale

        sc, sh = bn_scale("encoder.0.bn")
        w0 = g("encoder.0.conv.weight").reshape(32, 27) * sc[:, None]
        pk["enc0_w"], pk["enc0_b"] = _dev32(w0, dev), _dev32(g("encoder.0.conv.bias") * sc + sh, dev)
        for i in (1, 2, 3):
            sc, sh = bn_scale(f"encoder.{i}.bn")
            w = g(f"encoder.{i}.conv.weight") * sc[:, None, None, None, None]
            wk, _, kpad = _pack_conv(w, None)
            pk[f"enc{i}"] = {"w": _dev16(wk, dev), "b": _dev32(g(f"encoder.{i}.conv.bias") * sc + sh, dev), "kpad": kpad,
                             "cin": w.shape[1], "cout": w.shape[0], "k": 3}
        # Linear(2048, 512) reads the (c, z, y, x) flatten; activations here are (z, y, x, c)
        perm = np.arange(256 * 8).reshape(256, 8).T.reshape(-1)
        pk["lin_w"], pk["lin_b"] = _dev16(g("encoder.5.weight")[:, perm], dev), _dev32(g("encoder.5.bias"), dev)
        pk["fc_w"] = _dev16(np.concatenate([g("fc_mu.weight"), g("fc_logvar.weight")], 0), dev)
        pk["fc_b"] = _dev32(np.concatenate([g("fc_mu.bias"), g("fc_logvar.bias")], 0), dev)
        pk["din_w"], pk["din_b"] = _dev16(g("decoder_input.weight")[perm], dev), _dev32(g("decoder_input.bias")[perm], dev)
        for i in (0, 1, 2):
            sc, sh = bn_scale(f"decoder.{i}.bn")
            w = g(f"decoder.{i}.deconv.weight") * sc[None, :, None, None, None]
            classes = []
            for pz in (0, 1):
                for py in (0, 1):
                    for px in (0, 1):
                        wc, taps, kpad = _pack_convT3_class(w, pz, py, px)
                        classes.append({"w": _dev16(wc, dev), "taps": torch.from_numpy(taps).to(dev), "kpad": kpad,
                                        "p": (pz, py, px)})
            pk[f"dec{i}"] = {"classes": classes, "b": _dev32(g(f"decoder.{i}.deconv.bias") * sc + sh, dev),
                             "cin": w.shape[0], "cout": w.shape[1]}
        wl = g("decoder.3.weight")                                   # (32, 1, 3, 3, 3)
        pk["last_w"] = _dev32(np.transpose(wl[:, 0], (1, 2, 3, 0)).reshape(27, 32), dev)
        pk["last_b"] = float(g("decoder.3.bias")[0])
        self._packed = pk
        return pk

    def _conv_s2(self, L, x, b, din):
        lib = _lib.load()
        dout = din // 2
        out = torch.empty(b * dout ** 3, L["cout"], dtype=torch.float16, device=x.device)
        d = _lib.Conv3dDesc()
        d.inp, d.batch, d.in_d, d.in_h, d.in_w, d.cin = x.data_ptr(), b, din, din, din, L["cin"]
        d.rows_d = d.rows_h = d.rows_w = dout
        d.stride = 2
        taps = self._packed["taps3"]
        d.taps, d.ntaps, d.kpad = taps.data_ptr(), 27, L["kpad"]
        d.w, d.bias, d.resid, d.relu = L["w"].data_ptr(), L["b"].data_ptr(), 0, 1
        d.out, d.cout = out.data_ptr(), L["cout"]
        d.out_d = d.out_h = d.out_w = dout
        d.out_scale = 1
        d.zero_page = self._packed["zero"].data_ptr()
        _launch_convs([d], x.device, "conv3d")
        return out

    def _deconv(self, L, x, b, din):
        dout = 2 * din
        out = torch.empty(b * dout ** 3, L["cout"], dtype=torch.float16, device=x.device)
        groups: Dict[int, list] = {}                                # classes with the same tap count share a launch
        for cls in L["classes"]:
            d = _lib.Conv3dDesc()
            d.inp, d.batch, d.in_d, d.in_h, d.in_w, d.cin = x.data_ptr(), b, din, din, din, L["cin"]
            d.rows_d = d.rows_h = d.rows_w = din
            d.stride = 1
            d.taps, d.ntaps, d.kpad = cls["taps"].data_ptr(), cls["taps"].numel(), cls["kpad"]
            d.w, d.bias, d.resid, d.relu = cls["w"].data_ptr(), L["b"].data_ptr(), 0, 1
            d.out, d.cout = out.data_ptr(), L["cout"]
            d.out_d = d.out_h = d.out_w = dout
            d.out_scale = 2
            d.out_off_z, d.out_off_y, d.out_off_x = cls["p"]
            d.zero_page = self._packed["zero"].data_ptr()
            groups.setdefault(cls["kpad"] * 64 + int(cls["taps"].numel()), []).append(d)
        for descs in groups.values():
            _launch_convs(descs, x.device, "conv3d(T)")
        return out

    def encode(self, x: torch.Tensor):
        """networks.py:2044-2057."""
        from . import ops
        self._need_cuda(x)
        pk = self._ensure_packed()
        lib = _lib.load()
        b = x.shape[0]
        x = x.to(torch.float32).contiguous()
        h = torch.empty(b * 16 ** 3, 32, dtype=torch.float16, device=x.device)
        _lib.check(lib.pcd_conv3d_first(x.data_ptr(), b, 32, 32, 32, 2, pk["enc0_w"].data_ptr(), pk["enc0_b"].data_ptr(),
                                        32, h.data_ptr(), _lib.stream_ptr()), "conv3d_first")
        h = self._conv_s2(pk["enc1"], h, b, 16)
        h = self._conv_s2(pk["enc2"], h, b, 8)
        h = self._conv_s2(pk["enc3"], h, b, 4)                        # (B*8, 256) = (B, 2,2,2, 256)
        h = ops.gemm_f16(h.reshape(b, 2048), pk["lin_w"], pk["lin_b"], relu=True)
        out = ops.gemm_f16_out32(h, pk["fc_w"], pk["fc_b"])
        return out[:, :self.latent_dim].contiguous(), out[:, self.latent_dim:].contiguous()

    reparameterize = VAE3DLarge.reparameterize

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """networks.py:2077-2090."""
        from . import ops
        self._need_cuda(z)
        pk = self._ensure_packed()
        b = z.shape[0]
        h = ops.gemm_f16(z.to(torch.float16).contiguous(), pk["din_w"], pk["din_b"]).reshape(b * 8, 256)
        h = self._deconv(pk["dec0"], h, b, 2)
        h = self._deconv(pk["dec1"], h, b, 4)
        h = self._deconv(pk["dec2"], h, b, 8)                          # (B, 16,16,16, 32)
        out = torch.empty(b, 1, 32, 32, 32, dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().pcd_convt3d_last_sigmoid(h.data_ptr(), b, 16, 16, 16, 32, pk["last_w"].data_ptr(),
                                                        pk["last_b"], out.data_ptr(), _lib.stream_ptr()), "convT3d_last")
        return out

    forward = VAE3DLarge.forward
    sample = VAE3DLarge.sample
