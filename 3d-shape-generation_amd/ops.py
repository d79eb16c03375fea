"""Thin tensor-level wrappers over the layer-level C entry points (used by the network
modules and by the parity tests).  Tensors are torch CUDA tensors used as device buffers."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _desc(a1, w, bias, a2=None, shape_bias=None, rows_per_shape=0, relu=False, m=None) -> _lib.GemmDesc:
    d = _lib.GemmDesc()
    assert a1.dtype == torch.float16 and w.dtype == torch.float16 and a1.is_contiguous() and w.is_contiguous()
    d.a1, d.lda1, d.k1 = a1.data_ptr(), a1.shape[-1], a1.shape[-1]
    if a2 is not None:
        assert a2.dtype == torch.float16 and a2.is_contiguous()
        d.a2, d.lda2, d.k2 = a2.data_ptr(), a2.shape[-1], a2.shape[-1]
    d.w, d.ldw = w.data_ptr(), w.shape[1]
    d.bias = _lib.ptr(bias)
    d.shape_bias, d.rows_per_shape = _lib.ptr(shape_bias), rows_per_shape
    d.relu = 1 if relu else 0
    d.m = a1.numel() // a1.shape[-1] if m is None else m
    d.c = w.shape[0]
    return d


def gemm_f16(a1, w, bias=None, a2=None, shape_bias=None, rows_per_shape=0, relu=False,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = _desc(a1, w, bias, a2, shape_bias, rows_per_shape, relu)
    if out is None:
        out = torch.empty(d.m, d.c, dtype=torch.float16, device=a1.device)
    _lib.check(_lib.load().pcd_gemm_f16(d, out.data_ptr(), d.c, _lib.stream_ptr()), "gemm_f16")
    return out


def gemm_f16_hilo(a1, w_hilo, bias=None, a2=None, relu=False) -> torch.Tensor:
    """out = act([a1 | a2] . (hi + lo)^T + bias): `w_hilo` is [C][2 K] = the fp16 weights | the fp16 of their rounding residuals."""
    d = _desc(a1, w_hilo, bias, a2, relu=relu)
    assert w_hilo.shape[1] == 2 * (d.k1 + d.k2)
    out = torch.empty(d.m, d.c, dtype=torch.float16, device=a1.device)
    _lib.check(_lib.load().pcd_gemm_f16_hilo(d, out.data_ptr(), d.c, _lib.stream_ptr()), "gemm_f16_hilo")
    return out


def gemm_f16_out32(a1, w, bias=None, a2=None, shape_bias=None, rows_per_shape=0, relu=False) -> torch.Tensor:
    d = _desc(a1, w, bias, a2, shape_bias, rows_per_shape, relu)
    out = torch.empty(d.m, d.c, dtype=torch.float32, device=a1.device)
    _lib.check(_lib.load().pcd_gemm_f16_out32(d, out.data_ptr(), d.c, _lib.stream_ptr()), "gemm_f16_out32")
    return out


def gemm_f16_residual(a1, w, bias, resid, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = _desc(a1, w, bias)
    assert resid.dtype == torch.float16 and resid.is_contiguous() and resid.shape[-1] == d.c
    if out is None:
        out = torch.empty(d.m, d.c, dtype=torch.float16, device=a1.device)
    _lib.check(_lib.load().pcd_gemm_f16_residual(d, resid.data_ptr(), d.c, out.data_ptr(), d.c, _lib.stream_ptr()),
               "gemm_f16_residual")
    return out


def gemm_f16_colmax(a1, w, bias, rows_per_shape: int) -> torch.Tensor:
    d = _desc(a1, w, bias, relu=True)
    n_shapes = (d.m + rows_per_shape - 1) // rows_per_shape
    out = torch.empty(n_shapes, d.c, dtype=torch.float32, device=a1.device)
    lib = _lib.load()
    _lib.check(lib.pcd_fill_zero(out.data_ptr(), out.numel() * 4, _lib.stream_ptr()), "fill_zero")
    _lib.check(lib.pcd_gemm_f16_colmax(d, out.data_ptr(), rows_per_shape, _lib.stream_ptr()), "gemm_f16_colmax")
    return out


def layernorm_f16(x, gamma, beta) -> torch.Tensor:
    assert x.dtype == torch.float16 and x.is_contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().pcd_layernorm_f16(x.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], gamma.data_ptr(),
                                             beta.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "layernorm")
    return out


def set_attention_f16(qkv, batch: int, n_points: int, c: int, heads: int) -> torch.Tensor:
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and qkv.shape[-1] == 3 * c
    lib = _lib.load()
    nws = lib.pcd_set_attention_workspace_bytes(batch, n_points, c)      # 0 since the transpose-read kernel
    ws = torch.empty(nws, dtype=torch.uint8, device=qkv.device) if nws else None
    out = torch.empty(batch * n_points, c, dtype=torch.float16, device=qkv.device)
    _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), batch, n_points, c, heads, out.data_ptr(), _lib.ptr(ws),
                                         nws, _lib.stream_ptr()), "set_attention")
    return out


def linear_f32(x, w, b) -> torch.Tensor:
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().pcd_linear_f32(x.data_ptr(), x.shape[0], x.shape[1], w.data_ptr(), _lib.ptr(b), w.shape[0],
                                          out.data_ptr(), _lib.stream_ptr()), "linear_f32")
    return out
