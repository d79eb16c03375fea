"""Denoiser / VAE modules with the reference's constructor signatures and `state_dict`
contract (reference networks.py), executing on hand-written gfx950 kernels.

The modules are parameter containers (a `ParamTree` generated from `specs`) plus a
packed device-side weight cache; `forward` enqueues HIP kernels through the C ABI
(`_lib`).  There is no PyTorch compute path and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
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


# ------------------------------------------------------------------ UNetPointNetLarge
class UNetPointNetLarge(_HipModule):
    """Drop-in for reference networks.py:724-838: eps = model(x (B,N,3), t (B,))."""

    def __init__(self, dim: int = 512, time_dim: int = 256):
        super().__init__()
        self.dim, self.time_dim = dim, time_dim
        self._build_from_spec(specs.unet_pointnet_large_spec(dim, time_dim))
        self._handle = None

    # -- packing -----------------------------------------------------------------
    def _release(self):
        if getattr(self, "_handle", None):
            _lib.load().pcd_unet_destroy(self._handle)
        self._handle = None

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
        lin, ex = packing.pack_point_unet(self.state_dict(), "", self.time_dim, self.dim)
        keep = {"freqs": packing.timestep_freqs(self.time_dim).to(dev)}
        for k in ("tw0", "tb0", "tw2", "tb2", "e1w_xyz", "e1w_t", "e1b", "head_w", "head_b"):
            keep[k] = _dev32(ex[k], dev)
        keep["wg"] = _dev16(ex["wg"], dev)
        desc = _lib.UnetDesc()
        desc.time_dim, desc.dim = self.time_dim, self.dim
        for k in ("freqs", "tw0", "tb0", "tw2", "tb2", "e1w_xyz", "e1w_t", "e1b", "head_w", "head_b", "wg"):
            setattr(desc, k, keep[k].data_ptr())
        desc.wg_k, desc.wg_c = 4096, 1024
        for i, (w, b) in enumerate(lin):
            keep[f"w{i}"], keep[f"b{i}"] = _dev16(w, dev), _dev32(b, dev)
            desc.lin[i].w, desc.lin[i].b = keep[f"w{i}"].data_ptr(), keep[f"b{i}"].data_ptr()
            desc.lin[i].c, desc.lin[i].k = w.shape
        handle = C.c_void_p()
        _lib.check(lib.pcd_unet_create(C.byref(desc), C.byref(handle)), "unet_create")
        self._handle = handle
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
        nbytes = lib.pcd_unet_workspace_bytes(b, n)
        ws = self._workspace((b, n), nbytes)
        if out is None:
            out = torch.empty_like(x)
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

    def tap(self, name: str, batch: int, n_points: int) -> torch.Tensor:
        """Intermediate of the last forward (parity tests): x1..x4 (B,N,C) fp16, pooled/gbias fp32."""
        widths = {"x1": 128, "x2": 256, "x3": 512, "x4": 1024}
        ws = self._ws[(batch, n_points)]
        if name in widths:
            dst = torch.empty(batch, n_points, widths[name], dtype=torch.float16, device=self.device)
        else:
            dst = torch.empty(batch, {"pooled": 4096, "gbias": 1024}[name], dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().pcd_unet_tap(self._handle, name.encode(), batch, n_points, ws.data_ptr(),
                                            dst.data_ptr(), dst.numel() * dst.element_size(), _lib.stream_ptr()), "tap")
        return dst
