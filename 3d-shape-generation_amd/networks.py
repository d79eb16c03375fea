"""Denoiser / VAE modules with the reference's constructor signatures and `state_dict`
contract (reference networks.py), executing on hand-written gfx950 kernels.

The modules are parameter containers (a `ParamTree` generated from `specs`) plus a
packed device-side weight cache; `forward` enqueues HIP kernels through the C ABI
(`_lib`).  There is no PyTorch compute path and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, packing, specs


# --------------------------------------------------------------------------- params
class _Node(nn.Module):
    pass


class ParamTree(nn.Module):
    """nn.Module whose parameters/buffers are generated from a spec list so that
    `state_dict()` has exactly the reference's keys, shapes and order."""

    def _build_from_spec(self, spec: specs.Spec) -> None:
        for key, shape, role in spec:
            parts = key.split(".")
            node: nn.Module = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            leaf = parts[-1]
            if role in ("w", "wT"):
                t = torch.empty(shape)
                # kaiming-normal fan_out (reference diffusion.py:44-48); fan_out = shape[0] * receptive field
                fan_out = shape[0] * int(np.prod(shape[2:]))   # torch's definition, transposed convs included
                t.normal_(0.0, math.sqrt(2.0 / max(fan_out, 1)))
                node.register_parameter(leaf, nn.Parameter(t))
            elif role in ("b", "beta"):
                node.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
            elif role == "g":
                node.register_parameter(leaf, nn.Parameter(torch.ones(shape)))
            elif role == "rm":
                node.register_buffer(leaf, torch.zeros(shape))
            elif role == "rv":
                node.register_buffer(leaf, torch.ones(shape))
            elif role == "nbt":
                node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
            else:
                raise ValueError(role)


class _HipModule(ParamTree):
    """Common machinery: packed-weight cache invalidation and device checks."""

    def __init__(self):
        super().__init__()
        self._packed = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        # a parent's load_state_dict recurses through _load_from_state_dict and never calls a child's
        # load_state_dict override: the post hook fires for every module that received keys
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self) -> None:
        """Drop the packed device weights (call after mutating parameters in place)."""
        self._release()
        self._packed = None
        self._ws = {}

    def _release(self) -> None:
        pass

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _need_cuda(self, *tensors) -> torch.device:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs only on an MI355X device; call .to('cuda') "
                               "(this framework has no CPU path)")
        for t in tensors:
            if t is not None and t.device != dev:
                raise RuntimeError(f"input on {t.device}, module on {dev}")
        return dev

    def _workspace(self, key: tuple, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws = {key: ws}   # keep only the latest shape's workspace
        return ws


def _dev16(a: np.ndarray, dev) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float16).to(dev).contiguous()


def _dev32(a: np.ndarray, dev) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32).to(dev).contiguous()


PRECISIONS = ("fp16", "fp32")


def _check_precision(precision: str) -> None:
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")


def _env_precision() -> str:
    """PCD_PARITY=fp32 puts every denoiser created afterwards into its fp32 parity mode."""
    p = os.environ.get("PCD_PARITY", "fp16")
    if p not in PRECISIONS:
        raise ValueError(f"PCD_PARITY must be one of {PRECISIONS}, got {p!r}")
    return p


# ------------------------------------------------------------------ UNetPointNetLarge
class UNetPointNetLarge(_HipModule):
    """Drop-in for reference networks.py:724-838: eps = model(x (B,N,3), t (B,)).

    `precision`: "fp16" (default; fp16 operands on the fp16 matrix cores, fp32 accumulation: csrc/unet.hip) or "fp32"
    (SURVEY 8(c)'s parity mode: fp32 weights, activations and products, csrc/unet_f32.hip -- the reference's own
    arithmetic type, held to eps rel-L2 <= 1e-4).  Chosen per module with `set_precision`, or for every module created
    afterwards with the environment variable PCD_PARITY=fp32.  The constructor signature stays the reference's."""

    PRECISIONS = ("fp16", "fp32")

    def __init__(self, dim: int = 512, time_dim: int = 256):
        super().__init__()
        self.dim, self.time_dim = dim, time_dim
        self._build_from_spec(specs.unet_pointnet_large_spec(dim, time_dim))
        self._handle = None
        self._handle_f32 = False
        self._capture = None
        self.precision = os.environ.get("PCD_PARITY", "fp16")
        if self.precision not in self.PRECISIONS:
            raise ValueError(f"PCD_PARITY must be one of {self.PRECISIONS}, got {self.precision!r}")
        # fp16 mode: the narrow layers at the two ends of the U-net (enc1.conv2/3, enc2.conv3, dec1.*, output.0: the direct route from
        # the coordinates to the predicted noise) carry hi / lo weights -- the fp16 weights and the fp16 of their rounding residuals,
        # two MFMA passes, < 2 % of the FLOPs -- because the fp16 rounding of THESE weights is what the 1000-step DDPM trajectory
        # deviated by.  PCD_NARROW_HILO=0 / set_hilo_mask(0): plain fp16 weights everywhere.
        self.hilo_mask = _lib.PCD_UNET_HILO_ALLOWED if os.environ.get("PCD_NARROW_HILO", "1") != "0" else 0

    def set_hilo_mask(self, mask: int) -> "UNetPointNetLarge":
        if mask & ~_lib.PCD_UNET_HILO_ALLOWED:
            raise ValueError(f"hi / lo weights are available for layers {_lib.PCD_UNET_HILO_ALLOWED:#x} only, got {mask:#x}")
        if mask != self.hilo_mask:
            self.invalidate()
            self.hilo_mask = mask
        return self

    def set_precision(self, precision: str) -> "UNetPointNetLarge":
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}, got {precision!r}")
        if precision != self.precision:
            self.invalidate()
            self.precision = precision
        return self

    # -- packing -----------------------------------------------------------------
    def _release(self):
        if getattr(self, "_handle", None):
            lib = _lib.load()
            (lib.pcd_unet_f32_destroy if self._handle_f32 else lib.pcd_unet_destroy)(self._handle)
        self._handle = None
        self._capture = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        dev = self._need_cuda()
        _lib.require_gpu()
        lib = _lib.load()
        f32 = self.precision == "fp32"
        devw = _dev32 if f32 else _dev16
        lin, ex = packing.pack_point_unet(self.state_dict(), "", self.time_dim, self.dim)
        keep = {"freqs": packing.timestep_freqs(self.time_dim).to(dev)}
        for k in ("tw0", "tb0", "tw2", "tb2", "e1w_xyz", "e1w_t", "e1b", "head_w", "head_b"):
            keep[k] = _dev32(ex[k], dev)
        keep["wg"] = devw(ex["wg"], dev)
        desc = _lib.UnetDesc()
        desc.time_dim, desc.dim = self.time_dim, self.dim
        for k in ("freqs", "tw0", "tb0", "tw2", "tb2", "e1w_xyz", "e1w_t", "e1b", "head_w", "head_b", "wg"):
            setattr(desc, k, keep[k].data_ptr())
        desc.wg_k, desc.wg_c = 4096, 1024
        desc.hilo_mask = 0 if f32 else self.hilo_mask
        for i, (w, b) in enumerate(lin):
            if (desc.hilo_mask >> i) & 1:
                keep[f"w{i}"] = _dev16(packing.split_hilo(w), dev)                   # [C][2 K] = hi | lo
            else:
                keep[f"w{i}"] = devw(w, dev)
            keep[f"b{i}"] = _dev32(b, dev)
            desc.lin[i].w, desc.lin[i].b = keep[f"w{i}"].data_ptr(), keep[f"b{i}"].data_ptr()
            desc.lin[i].c, desc.lin[i].k = w.shape
        handle = C.c_void_p()
        if f32:
            _lib.check(lib.pcd_unet_f32_create(C.byref(desc), C.byref(handle)), "unet_f32_create")
        else:
            _lib.check(lib.pcd_unet_create(C.byref(desc), C.byref(handle)), "unet_create")
        self._handle, self._handle_f32 = handle, f32
        self._packed = keep
        return keep

    # -- pieces used by the samplers ---------------------------------------------------
    def time_bias(self, t: torch.Tensor) -> torch.Tensor:
        """Hoisted time half of enc1.conv1 for each value of t: (len(t), 64) fp32 (K3)."""
        pk = self._ensure_packed()
        t = t.to(self.device, torch.float32).contiguous()
        out = torch.empty(t.numel(), 64, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().pcd_time_embed(
            t.data_ptr(), t.numel(), pk["freqs"].data_ptr(), self.time_dim, self.dim,
            pk["tw0"].data_ptr(), pk["tb0"].data_ptr(), pk["tw2"].data_ptr(), pk["tb2"].data_ptr(),
            0, pk["e1w_t"].data_ptr(), pk["e1b"].data_ptr(), 64, out.data_ptr(), _lib.stream_ptr()), "time_embed")
        return out

    def time_mlp_out(self, t: torch.Tensor) -> torch.Tensor:
        """time_mlp(get_timestep_embedding(t)) (networks.py:791-792), for parity tests."""
        pk = self._ensure_packed()
        t = t.to(self.device, torch.float32).contiguous()
        out = torch.empty(t.numel(), self.dim, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().pcd_time_embed(
            t.data_ptr(), t.numel(), pk["freqs"].data_ptr(), self.time_dim, self.dim,
            pk["tw0"].data_ptr(), pk["tb0"].data_ptr(), pk["tw2"].data_ptr(), pk["tb2"].data_ptr(),
            out.data_ptr(), 0, 0, 0, 0, _lib.stream_ptr()), "time_embed")
        return out

    def forward_with_bias(self, x: torch.Tensor, tbias: torch.Tensor, shape_stride: int,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eps for x (B,N,3) given a precomputed time bias row (stride 0) or one row per shape (1)."""
        self._ensure_packed()
        b, n, _ = x.shape
        lib = _lib.load()
        if out is None:
            out = torch.empty_like(x)
        if self._handle_f32:
            ws = self._workspace((b, n), lib.pcd_unet_f32_workspace_bytes(b, n))
            _lib.check(lib.pcd_unet_f32_forward(self._handle, x.data_ptr(), b, n, tbias.data_ptr(), shape_stride,
                                                out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "unet_f32_forward")
            return out
        ws = self._workspace((b, n), lib.pcd_unet_workspace_bytes(b, n))
        _lib.check(lib.pcd_unet_forward(self._handle, x.data_ptr(), b, n, tbias.data_ptr(), shape_stride,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "unet_forward")
        return out

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        self._need_cuda(x, t)
        if x.dim() != 3 or x.shape[2] != 3:
            raise ValueError(f"x must be (B, N, 3), got {tuple(x.shape)}")
        if t.dim() != 1 or t.shape[0] != x.shape[0]:
            raise ValueError(f"t must be (B,), got {tuple(t.shape)} for batch {x.shape[0]}")
        x = x.to(torch.float32).contiguous()
        return self.forward_with_bias(x, self.time_bias(t), 1)

    _TAP_WIDTHS = {"x1": 128, "x2": 256, "x3": 512, "x4": 1024, "d4": 512, "d3": 256, "d2": 128, "d1": 64}

    def capture_decoder(self, batch: int, n_points: int, on: bool = True) -> None:
        """fp16 path, parity tests: make the following forwards of this shape keep the decoder blocks' outputs
        (dec4..dec1, networks.py:811-814) for `tap("d4".."d1")`; they otherwise live in ping-pong buffers and inside
        the chained tail.  The fp32 path always keeps them."""
        self._ensure_packed()
        if self._handle_f32:
            return
        if on:
            self._capture = {k: torch.empty(batch, n_points, self._TAP_WIDTHS[k], dtype=torch.float16, device=self.device)
                             for k in ("d4", "d3", "d2", "d1")}
            ptrs = [self._capture[k].data_ptr() for k in ("d4", "d3", "d2", "d1")]
        else:
            self._capture, ptrs = None, [0, 0, 0, 0]
        _lib.check(_lib.load().pcd_unet_capture(self._handle, *ptrs), "unet_capture")

    def tap(self, name: str, batch: int, n_points: int) -> torch.Tensor:
        """Intermediate of the last forward (parity tests): x1..x4 / d4..d1 as (B,N,C) (fp16, or fp32 in the fp32 mode),
        pooled / gbias fp32."""
        ws = self._ws[(batch, n_points)]
        lib = _lib.load()
        if name in self._TAP_WIDTHS:
            if not self._handle_f32 and name.startswith("d"):
                if self._capture is None or tuple(self._capture[name].shape[:2]) != (batch, n_points):
                    raise RuntimeError("decoder taps of the fp16 path need capture_decoder(batch, n_points) before the forward")
                return self._capture[name].clone()
            dst = torch.empty(batch, n_points, self._TAP_WIDTHS[name], device=self.device,
                              dtype=torch.float32 if self._handle_f32 else torch.float16)
        else:
            dst = torch.empty(batch, {"pooled": 4096, "gbias": 1024}[name], dtype=torch.float32, device=self.device)
        fn = lib.pcd_unet_f32_tap if self._handle_f32 else lib.pcd_unet_tap
        _lib.check(fn(self._handle, name.encode(), batch, n_points, ws.data_ptr(),
                      dst.data_ptr(), dst.numel() * dst.element_size(), _lib.stream_ptr()), "tap")
        return dst


# ------------------------------------------------------------------ set attention
class _PackedSAB:
    """Device weights of one SetAttentionBlock (reference networks.py:51-68) and their C descriptor."""

    def __init__(self, sd, prefix: str, dim: int, dev, f32: bool = False):
        g = lambda k: sd[prefix + k].detach().to("cpu", torch.float64).numpy()
        devw = _dev32 if f32 else _dev16                                   # fp32 parity mode: fp32 weights (csrc/attn_f32.hip)
        self.dim = dim
        self.w_in, self.b_in = devw(g("attention.in_proj_weight"), dev), _dev32(g("attention.in_proj_bias"), dev)
        self.w_out, self.b_out = devw(g("attention.out_proj.weight"), dev), _dev32(g("attention.out_proj.bias"), dev)
        self.ln1_g, self.ln1_b = _dev32(g("ln1.weight"), dev), _dev32(g("ln1.bias"), dev)
        self.ln2_g, self.ln2_b = _dev32(g("ln2.weight"), dev), _dev32(g("ln2.bias"), dev)
        self.w_ff1, self.b_ff1 = devw(g("ff.0.weight"), dev), _dev32(g("ff.0.bias"), dev)
        self.w_ff2, self.b_ff2 = devw(g("ff.2.weight"), dev), _dev32(g("ff.2.bias"), dev)
        self.tail = None
        self.lnlin = None
        self.ffn = None
        self.f32 = f32

    def fill(self, d: "_lib.SabDesc") -> "_lib.SabDesc":
        d.dim = self.dim
        for k in ("w_in", "b_in", "w_out", "b_out", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "w_ff1", "b_ff1", "w_ff2", "b_ff2"):
            setattr(d, k, getattr(self, k).data_ptr())
        d.tail_packed = None
        d.ln_in_packed = d.ln_ff1_packed = d.ffn_packed = None
        packed_now = False
        # C <= 128, fp16: the block's tail (out_proj + residual + LN2 + FFN + residual) as one launch needs its weights in fragment-order stage images
        lib = _lib.load()
        nbytes = 0 if self.f32 else lib.pcd_sab_tail_packed_bytes(self.dim)
        if nbytes:
            if self.tail is None:
                self.tail = torch.empty(nbytes, dtype=torch.uint8, device=self.w_out.device)
                _lib.check(lib.pcd_sab_tail_pack(C.byref(d), self.tail.data_ptr(), _lib.stream_ptr()), "sab_tail_pack")
                packed_now = True
            d.tail_packed = self.tail.data_ptr()
        # C = 256, fp16: LN1 + in_proj and LN2 + ff.0 as one launch each (wide-chain kernel with a LayerNorm prologue)
        if self.dim == 256 and not self.f32:
            if self.lnlin is None:
                bufs = []
                for w, b, passes, g_, b_ in ((self.w_in, self.b_in, 3, self.ln1_g, self.ln1_b), (self.w_ff1, self.b_ff1, 4, self.ln2_g, self.ln2_b)):
                    buf = torch.empty(lib.pcd_pw_wide_ln_linear_packed_bytes(passes), dtype=torch.uint8, device=self.w_out.device)
                    _lib.check(lib.pcd_pw_wide_ln_linear_pack(w.data_ptr(), b.data_ptr(), passes, g_.data_ptr(), b_.data_ptr(), buf.data_ptr(),
                                                              _lib.stream_ptr()), "pw_wide_ln_linear_pack")
                    bufs.append(buf)
                self.lnlin = bufs
                packed_now = True
            d.ln_in_packed, d.ln_ff1_packed = self.lnlin[0].data_ptr(), self.lnlin[1].data_ptr()
            # ... and LN2 + ff.0 + ReLU + ff.2 + residual as ONE launch (csrc/wideffn.hip): the 4C-wide hidden tensor is never written
            if self.ffn is None:
                self.ffn = torch.empty(lib.pcd_wide_ffn_packed_bytes(), dtype=torch.uint8, device=self.w_out.device)
                _lib.check(lib.pcd_wide_ffn_pack(self.w_ff1.data_ptr(), self.b_ff1.data_ptr(), self.w_ff2.data_ptr(), self.b_ff2.data_ptr(),
                                                 self.ln2_g.data_ptr(), self.ln2_b.data_ptr(), self.ffn.data_ptr(), _lib.stream_ptr()), "wide_ffn_pack")
                packed_now = True
            d.ffn_packed = self.ffn.data_ptr()
        if packed_now:
            # the images were written by pack kernels on the CURRENT stream; a forward enqueued on another stream (a side stream, a capture) must find them
            # complete: one synchronisation per model packing
            torch.cuda.current_stream().synchronize()
        return d


class SetAttentionBlock(_HipModule):
    """Drop-in for reference networks.py:51-83: (B, N, C) -> (B, N, C), pre-LN MHA + FFN, one pcd_sab_forward call."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self._build_from_spec(specs.set_attention_spec(dim))
        self.precision = _env_precision()

    def set_precision(self, precision: str) -> "SetAttentionBlock":
        """"fp16" (default: fp16 operands, flash-style kernel on the matrix cores) or "fp32" (the reference's arithmetic type, csrc/attn_f32.hip)."""
        _check_precision(precision)
        if precision != self.precision:
            self.invalidate()
            self.precision = precision
        return self

    def _ensure_packed(self):
        if self._packed is None:
            dev = self._need_cuda()
            _lib.require_gpu()
            pk = _PackedSAB(self.state_dict(), "", self.dim, dev, f32=self.precision == "fp32")
            self._packed = (pk, pk.fill(_lib.SabDesc()))
        return self._packed

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._need_cuda(x)
        b, n, c = x.shape
        if c != self.dim:
            raise ValueError(f"expected {self.dim} channels, got {c}")
        _, desc = self._ensure_packed()
        lib = _lib.load()
        if self.precision == "fp32":
            x32 = x.to(torch.float32).contiguous().reshape(b * n, c)
            y32 = torch.empty_like(x32)
            ws = self._workspace((b, n), lib.pcd_sab_f32_workspace_bytes(b * n, c))
            _lib.check(lib.pcd_sab_f32_forward(C.byref(desc), x32.data_ptr(), b, n, self.num_heads, y32.data_ptr(), ws.data_ptr(),
                                               ws.numel(), _lib.stream_ptr()), "sab_f32_forward")
            return y32.reshape(b, n, c).to(x.dtype)
        x16 = x.to(torch.float16).contiguous().reshape(b * n, c)
        y16 = torch.empty_like(x16)
        ws = self._workspace((b, n), lib.pcd_sab_workspace_bytes(b * n, c))
        _lib.check(lib.pcd_sab_forward(C.byref(desc), x16.data_ptr(), b, n, self.num_heads, y16.data_ptr(), ws.data_ptr(),
                                       ws.numel(), _lib.stream_ptr()), "sab_forward")
        return y16.reshape(b, n, c).to(x.dtype)


class UNetAttentionPointExperimental(_HipModule):
    """Drop-in for reference networks.py:597-722 (the carrier of the set-attention blocks): eps = model(x (B,N,3), t (B,)).
    One pcd_attn_unet_forward call per forward; `time_bias` / `forward_with_bias` give the samplers the same two-stage
    surface as UNetPointNetLarge (the whole time path once per sampler call, one 704-float row per timestep)."""

    TB = _lib.PCD_ATTN_UNET_TB

    def __init__(self, num_points, dim=256, num_heads=4, num_blocks=3, time_dim=256):
        super().__init__()
        self.num_points, self.dim, self.num_heads, self.time_dim = num_points, dim, num_heads, time_dim
        self._build_from_spec(specs.unet_attention_spec(dim, time_dim))
        self._handle = None
        self._handle32 = None
        self.precision = _env_precision()

    def set_precision(self, precision: str) -> "UNetAttentionPointExperimental":
        """"fp16" (default) or "fp32" (fp32 weights / activations / arithmetic, csrc/attn_f32.hip: the reference's arithmetic type, 1e-4)."""
        _check_precision(precision)
        if precision != self.precision:
            self.invalidate()
            self.precision = precision
        return self

    def _release(self):
        if getattr(self, "_handle", None):
            _lib.load().pcd_attn_unet_destroy(self._handle)
        if getattr(self, "_handle32", None):
            _lib.load().pcd_attn_unet_f32_destroy(self._handle32)
        self._handle = None
        self._handle32 = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        if self.dim != self.time_dim:
            raise RuntimeError("UNetAttentionPointExperimental needs dim == time_dim: emb* layers take time_dim "
                               "inputs but receive time_mlp's dim outputs (reference networks.py:613-624,664-668)")
        dev = self._need_cuda()
        _lib.require_gpu()
        sd = self.state_dict()
        g = lambda k: sd[k].detach().to("cpu", torch.float64).numpy()
        keep = {"freqs": packing.timestep_freqs(self.time_dim).to(dev)}
        desc = _lib.AttnUnetDesc()
        desc.dim, desc.time_dim, desc.heads = self.dim, self.time_dim, self.num_heads
        desc.freqs = keep["freqs"].data_ptr()
        for name, field in (("time_mlp.0", "t*0"), ("time_mlp.2", "t*2")):
            keep[name + ".w"], keep[name + ".b"] = _dev32(g(name + ".weight"), dev), _dev32(g(name + ".bias"), dev)
            setattr(desc, field.replace("*", "w"), keep[name + ".w"].data_ptr())
            setattr(desc, field.replace("*", "b"), keep[name + ".b"].data_ptr())
        for i, (name, _) in enumerate(specs.ATTN_UNET_EMB):
            keep[name + ".w"], keep[name + ".b"] = _dev32(g(name + ".weight"), dev), _dev32(g(name + ".bias"), dev)
            desc.emb_w[i], desc.emb_b[i] = keep[name + ".w"].data_ptr(), keep[name + ".b"].data_ptr()
        # enc1 = PointNetLayer(3, 64): the K=3 first conv runs in the xyz kernel
        w, b = packing.fold_conv_bn(sd, "enc1.conv1", "enc1.bn1")
        keep["e1w"], keep["e1b"] = _dev32(w, dev), _dev32(b, dev)
        desc.e1w, desc.e1b = keep["e1w"].data_ptr(), keep["e1b"].data_ptr()
        stages = [("enc1", 2), ("enc1", 3)] + [(n, i) for n in ("enc2", "enc3", "dec3", "dec2") for i in (1, 2, 3)]
        for j, (name, i) in enumerate(stages):
            w, b = packing.fold_conv_bn(sd, f"{name}.conv{i}", f"{name}.bn{i}")
            keep[f"w{j}"], keep[f"b{j}"] = _dev16(w, dev), _dev32(b, dev)
            desc.lin[j].w, desc.lin[j].b = keep[f"w{j}"].data_ptr(), keep[f"b{j}"].data_ptr()
            desc.lin[j].c, desc.lin[j].k = w.shape
        for j, (name, c) in enumerate((("att1", 64), ("att2", 128), ("att3", 256), ("bottleneck", 256), ("att_dec3", 256),
                                       ("att_dec2", 128), ("att_dec1", 64))):
            keep[name] = _PackedSAB(sd, name + ".", c, dev)
            keep[name].fill(desc.sab[j])
        # tail: dec1 = PointNetLayer(128, 3, 3) + output Conv1d(3,3)
        w1, b1 = packing.fold_conv_bn(sd, "dec1.conv1", "dec1.bn1")
        w2, b2 = packing.fold_conv_bn(sd, "dec1.conv2", "dec1.bn2")
        w3, b3 = packing.fold_conv_bn(sd, "dec1.conv3", "dec1.bn3")
        w4, b4 = packing.fold_conv_bn(sd, "output", None)
        keep["t_w1"], keep["t_b1"] = _dev32(w1, dev), _dev32(b1, dev)
        keep["t_w234"] = _dev32(np.stack([w2, w3, w4]), dev)
        keep["t_b234"] = _dev32(np.stack([b2, b3, b4]), dev)
        for k in ("t_w1", "t_b1", "t_w234", "t_b234"):
            setattr(desc, k, keep[k].data_ptr())
        handle = C.c_void_p()
        _lib.check(_lib.load().pcd_attn_unet_create(C.byref(desc), C.byref(handle)), "attn_unet_create")
        self._handle, self._packed = handle, keep
        if self.precision == "fp32":
            # the fp32 parity mode: a second descriptor whose layer / attention weights are fp32 (the time path above is fp32 in both modes
            # and stays with the first handle: pcd_attn_unet_time_bias)
            d32 = _lib.AttnUnetDesc()
            C.memmove(C.byref(d32), C.byref(desc), C.sizeof(desc))
            for j, (name, i) in enumerate(stages):
                w, b = packing.fold_conv_bn(sd, f"{name}.conv{i}", f"{name}.bn{i}")
                keep[f"w32_{j}"] = _dev32(w, dev)
                d32.lin[j].w = keep[f"w32_{j}"].data_ptr()
            for j, (name, c) in enumerate((("att1", 64), ("att2", 128), ("att3", 256), ("bottleneck", 256), ("att_dec3", 256),
                                           ("att_dec2", 128), ("att_dec1", 64))):
                keep[name + "_32"] = _PackedSAB(sd, name + ".", c, dev, f32=True)
                keep[name + "_32"].fill(d32.sab[j])
            h32 = C.c_void_p()
            _lib.check(_lib.load().pcd_attn_unet_f32_create(C.byref(d32), C.byref(h32)), "attn_unet_f32_create")
            self._handle32 = h32
        return keep

    def time_bias(self, t: torch.Tensor) -> torch.Tensor:
        """Every time-dependent vector of the network for each value of t: (len(t), 704) fp32 rows
        [enc1 bias | emb2 | emb3 | emb_dec3 | emb_dec2 | emb_dec1] (networks.py:664-698)."""
        self._ensure_packed()
        t = t.to(self.device, torch.float32).contiguous()
        out = torch.empty(t.numel(), self.TB, dtype=torch.float32, device=self.device)
        scratch = torch.empty(t.numel(), self.dim, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().pcd_attn_unet_time_bias(self._handle, t.data_ptr(), t.numel(), scratch.data_ptr(),
                                                       out.data_ptr(), _lib.stream_ptr()), "attn_unet_time_bias")
        return out

    def forward_with_bias(self, x: torch.Tensor, tbias: torch.Tensor, shape_stride: int,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._ensure_packed()
        b, n, _ = x.shape
        lib = _lib.load()
        if out is None:
            out = torch.empty_like(x)
        if self._handle32 is not None:
            ws = self._workspace((b, n), lib.pcd_attn_unet_f32_workspace_bytes(b, n))
            _lib.check(lib.pcd_attn_unet_f32_forward(self._handle32, x.data_ptr(), b, n, tbias.data_ptr(), shape_stride,
                                                     out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "attn_unet_f32_forward")
            return out
        ws = self._workspace((b, n), lib.pcd_attn_unet_workspace_bytes(b, n))
        _lib.check(lib.pcd_attn_unet_forward(self._handle, x.data_ptr(), b, n, tbias.data_ptr(), shape_stride,
                                             out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "attn_unet_forward")
        return out

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        self._need_cuda(x, t)
        if x.dim() != 3 or x.shape[2] != 3:
            raise ValueError(f"x must be (B, N, 3), got {tuple(x.shape)}")
        if t.dim() != 1 or t.shape[0] != x.shape[0]:
            raise ValueError(f"t must be (B,), got {tuple(t.shape)} for batch {x.shape[0]}")
        x = x.to(torch.float32).contiguous()
        return self.forward_with_bias(x, self.time_bias(t), 1)

    def tap(self, name: str, batch: int, n_points: int) -> torch.Tensor:
        """Skip tensor of the last forward (parity tests): x1 / x2 / x3 as (B, N, 64 / 128 / 256) fp16 (fp32 in the fp32 parity mode)."""
        ws = self._ws[(batch, n_points)]
        if self._handle32 is not None:
            dst = torch.empty(batch, n_points, {"x1": 64, "x2": 128, "x3": 256}[name], dtype=torch.float32, device=self.device)
            _lib.check(_lib.load().pcd_attn_unet_f32_tap(self._handle32, name.encode(), batch, n_points, ws.data_ptr(), dst.data_ptr(),
                                                         dst.numel() * 4, _lib.stream_ptr()), "attn_unet_f32_tap")
            return dst
        dst = torch.empty(batch, n_points, {"x1": 64, "x2": 128, "x3": 256}[name], dtype=torch.float16, device=self.device)
        _lib.check(_lib.load().pcd_attn_unet_tap(self._handle, name.encode(), batch, n_points, ws.data_ptr(), dst.data_ptr(),
                                                 dst.numel() * 2, _lib.stream_ptr()), "attn_unet_tap")
        return dst


# ------------------------------------------------------------------ latent denoiser
class SimpleLatentUNetPointNet(_HipModule):
    """Drop-in for reference networks.py:962-1106: eps = model(z (B, latent), t (B,)).
    Eval-mode only (Dropout(0.1) in dec1 is the identity, as in every sampler)."""

    def __init__(self, latent_dim, dim=512, time_dim=256, dropout_rate=0.1):
        super().__init__()
        self.latent_dim, self.dim, self.time_dim = latent_dim, dim, time_dim
        self._build_from_spec(specs.latent_unet_spec(latent_dim, dim, time_dim))
        self._handle = None
        self._handle_f32 = False
        self._persist = None
        self.precision = os.environ.get("PCD_PARITY", "fp16")     # "fp32": the parity mode of csrc/latent_f32.hip (see UNetPointNetLarge)
        if self.precision not in UNetPointNetLarge.PRECISIONS:
            raise ValueError(f"PCD_PARITY must be one of {UNetPointNetLarge.PRECISIONS}, got {self.precision!r}")

    def set_precision(self, precision: str) -> "SimpleLatentUNetPointNet":
        """"fp16" (default: fp16 operands, the per-layer launches / the persistent kernel) or "fp32" (fp32 weights, activations
        and products, one launch per layer: the reference's arithmetic type, eps rel-L2 <= 1e-4)."""
        if precision not in UNetPointNetLarge.PRECISIONS:
            raise ValueError(f"precision must be one of {UNetPointNetLarge.PRECISIONS}, got {precision!r}")
        if precision != self.precision:
            self.invalidate()
            self.precision = precision
        return self

    def _release(self):
        if getattr(self, "_handle", None):
            lib = _lib.load()
            (lib.pcd_latent_f32_destroy if self._handle_f32 else lib.pcd_latent_destroy)(self._handle)
        if getattr(self, "_persist", None):
            _lib.load().pcd_latent_persist_destroy(self._persist)
        self._handle = None
        self._persist = None
        self._persist_ws = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ the whole step as one persistent launch
    def persist_supported(self, batch: int) -> bool:
        """True when `csrc/latent_persist.hip` can run on this device for this batch (batch <= 64, a 256-CU gfx950)."""
        self._need_cuda()
        if self.precision != "fp16":
            return False                                   # the persistent kernel is an fp16-operand kernel
        return bool(_lib.load().pcd_latent_persist_supported(int(batch)))

    def _persist_handle(self):
        self._ensure_packed()
        if getattr(self, "_persist", None) is None:
            handle = C.c_void_p()
            _lib.check(_lib.load().pcd_latent_persist_create(C.byref(self._desc), C.byref(handle)), "latent_persist_create")
            self._persist = handle
        if getattr(self, "_persist_ws", None) is None:                   # its own buffer: the per-layer path keeps `_ws`
            self._persist_ws = torch.empty(_lib.load().pcd_latent_persist_workspace_bytes(self._persist), dtype=torch.uint8,
                                           device=self.device)
        return self._persist, self._persist_ws

    def persist_status(self) -> int:
        """Waits for the current stream (the one the launch was enqueued on) and returns the status word of the last persistent launch: 0 = every wait was met, else
        (wait kind << 16) | workgroup -- the launch was abandoned (0.2 s bound per wait) and its outputs are undefined."""
        h, ws = self._persist_handle()
        st = C.c_uint(0)
        _lib.check(_lib.load().pcd_latent_persist_status(ws.data_ptr(), C.byref(st), _lib.stream_ptr()), "latent_persist_status")
        return int(st.value)

    def check_persist_status(self):
        """Raises if the last persistent launch was abandoned (callers with no saved input to re-run from)."""
        st = self.persist_status()
        if st:
            raise RuntimeError(f"persistent latent kernel: wait kind {st >> 16} of workgroup {st & 0xffff} timed out "
                               "(is another process holding CUs of this GPU?  set PCD_LATENT_PERSISTENT=0 to use the per-layer launches)")

    def inject_persist_fault(self, workgroup: int, step: int = 0) -> None:
        """Recovery tests: workgroup `workgroup` of the following persistent launches leaves at step `step` (-1: off)."""
        h, _ = self._persist_handle()
        _lib.check(_lib.load().pcd_latent_persist_inject_fault(h, int(workgroup), int(step)), "latent_persist_inject_fault")

    def forward_persist(self, z, tbias_row, out=None):
        """eps = model(z, t) for ONE t shared by the batch (tbias_row = time_bias(t)[0], 128 values), one launch."""
        h, ws = self._persist_handle()
        if out is None:
            out = torch.empty_like(z)
        _lib.check(_lib.load().pcd_latent_persist_forward(h, z.data_ptr(), z.shape[0], tbias_row.data_ptr(), out.data_ptr(),
                                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "latent_persist_forward")
        return out

    def ddim_steps_persist(self, z, x0, bias_table, rates, counter, nsteps):
        """`nsteps` DDIM steps from table row counter[0] in ONE launch: z updated in place, x0 = the last step's x_0.
        rates (4, T, R) fp32 = (n, s, n_next, s_next)."""
        h, ws = self._persist_handle()
        _lib.check(_lib.load().pcd_latent_persist_ddim(h, z.data_ptr(), x0.data_ptr(), z.shape[0], bias_table.data_ptr(),
                                                       bias_table.shape[1], rates.data_ptr(), rates.shape[2], rates.shape[1],
                                                       counter.data_ptr(), int(nsteps), ws.data_ptr(), ws.numel(),
                                                       _lib.stream_ptr()), "latent_persist_ddim")

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        if (self.latent_dim, self.dim, self.time_dim) != (256, 512, 256):
            raise RuntimeError("the HIP latent denoiser is built for latent_dim=256, dim=512, time_dim=256 "
                               "(the configuration diffusion.py:362,380 instantiates)")
        dev = self._need_cuda()
        _lib.require_gpu()
        lin, gn, ex = packing.pack_latent_unet(self.state_dict(), "")
        keep = {"freqs": packing.timestep_freqs(self.time_dim).to(dev)}
        for k, v in ex.items():
            keep[k] = _dev32(v, dev)
        desc = _lib.LatentDesc()
        f32 = self.precision == "fp32"
        for i, (w, b) in enumerate(lin):
            keep[f"w{i}"], keep[f"b{i}"] = (_dev32 if f32 else _dev16)(w, dev), _dev32(b, dev)
            desc.lin[i].w, desc.lin[i].b = keep[f"w{i}"].data_ptr(), keep[f"b{i}"].data_ptr()
            desc.lin[i].c, desc.lin[i].k = w.shape
        for i, (gm, bt) in enumerate(gn):
            keep[f"g{i}"], keep[f"be{i}"] = _dev32(gm, dev), _dev32(bt, dev)
            desc.gn_gamma[i], desc.gn_beta[i] = keep[f"g{i}"].data_ptr(), keep[f"be{i}"].data_ptr()
        handle = C.c_void_p()
        if f32:
            _lib.check(_lib.load().pcd_latent_f32_create(C.byref(desc), C.byref(handle)), "latent_f32_create")
        else:
            _lib.check(_lib.load().pcd_latent_create(C.byref(desc), C.byref(handle)), "latent_create")
        self._handle, self._handle_f32, self._packed, self._desc = handle, f32, keep, desc
        return keep

    def time_bias(self, t: torch.Tensor) -> torch.Tensor:
        """Hoisted time half of enc1 for each value of t: (len(t), 128) fp32."""
        pk = self._ensure_packed()
        t = t.to(self.device, torch.float32).contiguous()
        out = torch.empty(t.numel(), 128, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().pcd_time_embed(
            t.data_ptr(), t.numel(), pk["freqs"].data_ptr(), self.time_dim, self.time_dim,
            pk["tw0"].data_ptr(), pk["tb0"].data_ptr(), pk["tw2"].data_ptr(), pk["tb2"].data_ptr(),
            0, pk["e1w_t"].data_ptr(), pk["e1b"].data_ptr(), 128, out.data_ptr(), _lib.stream_ptr()), "time_embed")
        return out

    def forward_with_bias(self, z, tbias, shape_stride, out=None):
        self._ensure_packed()
        lib = _lib.load()
        b = z.shape[0]
        if out is None:
            out = torch.empty_like(z)
        if self._handle_f32:
            ws = self._workspace((b,), lib.pcd_latent_f32_workspace_bytes(b))
            _lib.check(lib.pcd_latent_f32_forward(self._handle, z.data_ptr(), b, tbias.data_ptr(), shape_stride,
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "latent_f32_forward")
            return out
        ws = self._workspace((b,), lib.pcd_latent_workspace_bytes(b))
        _lib.check(lib.pcd_latent_forward(self._handle, z.data_ptr(), b, tbias.data_ptr(), shape_stride,
                                          out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "latent_forward")
        return out

    def forward(self, z: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        self._need_cuda(z, t)
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise ValueError(f"z must be (B, {self.latent_dim}), got {tuple(z.shape)}")
        z = z.to(torch.float32).contiguous()
        return self.forward_with_bias(z, self.time_bias(t), 1)


def __getattr__(name):   # VAE3DLarge lives in vae.py; keep `networks.VAE3DLarge` importable like the reference
    if name in ("VAE3DLarge", "VAE3D"):
        from . import vae
        return getattr(vae, name)
    raise AttributeError(name)
