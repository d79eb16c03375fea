"""Training step of the point denoiser on the HIP kernels.

Reference: `PointCloudDiffusion.training_step / diffusion_loss / configure_optimizers` (diffusion.py:56-86,
170-186): t ~ U(0,1) per shape, x_t = s x_0 + n eps, eps_hat = model(x_t, t) with the model in train() mode
(BatchNorm1d uses batch statistics and updates its running estimates, networks.py:31-48), loss =
F.l1_loss(eps, eps_hat), optimizer AdamW(lr, weight_decay=1e-5).

What runs where: every dense product (forward z = a W^T, backward-data da = dz W, backward-weight
dW = dz^T a) is the fp16 MFMA GEMM (`pcd_gemm_f16*`), fp32 accumulation; BatchNorm forward/backward, the
max-pool argmax/scatter, reductions, transposes, the loss and AdamW are the kernels of csrc/train.hip.  This
module is the host-side sequencing (what Lightning's `trainer.fit` + autograd do for the reference); torch is
used for buffers and for re-slicing weights.  Master weights, statistics, parameter gradients and the optimizer
state are fp32; activations and activation gradients are fp16 with a static loss scale.

Unlike the sampler there is no algebraic folding here: BatchNorm statistics depend on the batch, so every
Conv1d output is materialised, including the (B, N, 4096) `global_feat` tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import contextlib

import torch

from . import _lib
from .packing import timestep_freqs
from .specs import POINT_DEC, POINT_ENC, POINT_REFINE

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _gemm_desc(a1, k1, lda1, a2, k2, lda2, w, ldw, bias, shape_bias, rps, m, c):
    g = _lib.GemmDesc()
    g.a1, g.lda1, g.k1 = a1, lda1, k1
    g.a2, g.lda2, g.k2 = a2, lda2, k2
    g.w, g.ldw = w, ldw
    g.bias = bias
    g.shape_bias, g.rows_per_shape = shape_bias, rps
    g.relu, g.m, g.c = 0, m, c
    return g


def _flatten_parameters(model, dev):
    """Re-point every trainable `nn.Parameter` of `model` into one flat fp32 buffer (the optimizer is then a single
    launch and `state_dict()` always shows the trained weights); returns (P, G, M1, M2, views, grad views)."""
    named = [(n, q) for n, q in model.named_parameters() if q.requires_grad]
    total = sum(q.numel() for _, q in named)
    P = torch.empty(total, dtype=torch.float32, device=dev)
    G, M1, M2 = torch.zeros_like(P), torch.zeros_like(P), torch.zeros_like(P)
    views: Dict[str, torch.Tensor] = {}
    grads: Dict[str, torch.Tensor] = {}
    off = 0
    for name, prm in named:
        n = prm.numel()
        view = P[off:off + n].view(prm.shape)
        view.copy_(prm.data)
        prm.data = view
        views[name] = view
        grads[name] = G[off:off + n].view(prm.shape)
        off += n
    return P, G, M1, M2, views, grads


def _allreduce_gradients(G: torch.Tensor) -> int:
    """Data-parallel training: SUM the flat gradient buffer over the ranks (one collective per step: RCCL all-reduce over
    xGMI under the `nccl` backend; staged through the host under `gloo`, which the CPU-side tests use).  Returns the
    world size; the caller folds the 1/world of the mean into the optimizer's gradient scale."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1
    if dist.get_backend() == "nccl":
        dist.all_reduce(G)
    else:
        h = G.cpu()
        dist.all_reduce(h)
        G.copy_(h)
    return dist.get_world_size()


def _collective_inplace(fn, t: torch.Tensor, *args) -> None:
    """Run an in-place collective (`dist.broadcast`, `dist.all_reduce`) on a device tensor: directly under `nccl` (RCCL),
    staged through the host under `gloo`."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" or t.device.type == "cpu":
        fn(t, *args)
    else:
        h = t.cpu()
        fn(h, *args)
        t.copy_(h)


class _Conv:
    """One Conv1d(k=1) [+ BatchNorm1d + ReLU]: names of its parameters and its saved tensors."""

    def __init__(self, conv: str, bn: Optional[str], cin: int, cout: int):
        self.conv, self.bn, self.cin, self.cout = conv, bn, cin, cout
        self.z = self.a = self.mean = self.var = None
        self.inputs: List[Tuple[torch.Tensor, int]] = []


class PointTrainer:
    """Forward + backward + AdamW for `UNetPointNetLarge` (networks.py:725-818).  Parameters stay the
    module's own `nn.Parameter`s (re-pointed into one flat fp32 buffer), so `state_dict()` always shows the
    trained weights and checkpoints keep the reference's keys."""

    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 loss_scale: float = 1024.0):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.model = model
        self.lr, self.wd, self.betas, self.eps, self.loss_scale = lr, weight_decay, betas, eps, float(loss_scale)
        self.dev = model.device
        if self.dev.type != "cuda":
            raise RuntimeError("PointTrainer needs the model on an MI355X (model.to('cuda'))")
        self.step_count = 0
        # ---- one flat fp32 buffer for parameters, one for gradients, two for the AdamW moments
        self.P, self.G, self.M1, self.M2, self.p, self.g = _flatten_parameters(model, self.dev)
        self.buf = dict(model.named_buffers())
        self.freqs = timestep_freqs(256)
        # ---- layer table in execution order
        self.enc = [[_Conv(f"{n}.conv{i}", f"{n}.bn{i}", a, b) for i, (a, b) in
                     enumerate([(259 if cin is None else cin, mid), (mid, mid), (mid, cout)], start=1)]
                    for n, cin, mid, cout in POINT_ENC]
        self.gf = [_Conv("global_feat.0", "global_feat.1", 1024, 2048), _Conv("global_feat.3", "global_feat.4", 2048, 4096)]
        self.dec = [[_Conv(f"{n}.conv{i}", f"{n}.bn{i}", a, b) for i, (a, b) in
                     enumerate([(cin, mid), (mid, mid), (mid, cout)], start=1)] for n, cin, mid, cout in POINT_DEC]
        self.out0 = _Conv("output.0", "output.1", 64, 64)
        self.refine = {c: _Conv(n, None, c, c) for n, c in POINT_REFINE}
        self.w16: Dict[str, torch.Tensor] = {}
        self.w16t: Dict[str, torch.Tensor] = {}
        self._ws: Dict[str, torch.Tensor] = {}
        self.debug: Optional[Dict[str, torch.Tensor]] = None    # tests set a dict: per-layer da / dz copies are kept
        self.refresh_weights()
        # load_state_dict copies into the flat buffer in place: the fp16 operand copies must follow
        model.register_load_state_dict_post_hook(lambda module, incompatible: self.refresh_weights())

    # ------------------------------------------------------------------ helpers
    def _st(self):
        return _lib.stream_ptr()

    def _chk(self, rc, what):
        _lib.check(rc, what)

    def _buf(self, key: str, shape, dtype) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    def _all_convs(self):
        for blk in self.enc:
            yield from blk
        yield from self.gf
        for blk in self.dec:
            yield from blk
        yield self.out0
        yield from self.refine.values()

    def refresh_weights(self):
        """fp16 operand copies of the fp32 master weights: W [C][K] for the forward product and W^T [K][C] for
        backward-data.  enc1.conv1 (K = 3 + 256) and the global half of dec4.conv1 stay fp32."""
        lib, st = self.lib, self._st()
        for L in self._all_convs():
            w = self.p[L.conv + ".weight"]
            if L.conv == "enc1.conv1":
                self.w_xyz = w[:, :3, 0].contiguous()
                continue
            if L.conv == "dec4.conv1":
                w2 = w[:, 4096:, 0].contiguous()               # the refine4(x4) half; the 4096 global columns stay fp32
            else:
                w2 = w.view(w.shape[0], w.shape[1])
            c, k = w2.shape
            a = self.w16.get(L.conv)
            if a is None:
                a = self.w16[L.conv] = torch.empty(c, k, dtype=torch.float16, device=self.dev)
                self.w16t[L.conv] = torch.empty(k, c, dtype=torch.float16, device=self.dev)
            self._chk(lib.pcd_f32_to_f16(w2.data_ptr(), a.data_ptr(), a.numel(), st), "f32_to_f16")
            self._chk(lib.pcd_transpose_f16(a.data_ptr(), c, k, self.w16t[L.conv].data_ptr(), st), "transpose")
        self.model.invalidate()        # the sampler's packed (BN-folded) weights are stale now

    def _mm(self, a, lda, ta, b, ldb, tb, m, n, k, bias, acc, c, ldc):
        self._chk(self.lib.pcd_matmul_f32(a, lda, ta, b, ldb, tb, m, n, k, bias, acc, c, ldc, self._st()), "matmul_f32")

    def _gemm(self, a1, k1, a2, k2, w, ldw, bias, shape_bias, rps, m, c, out, resid=None):
        g = _gemm_desc(a1.data_ptr(), k1, k1, a2.data_ptr() if a2 is not None else None, k2, k2, w, ldw,
                       bias, shape_bias, rps, m, c)
        if out.dtype == torch.float32:
            self._chk(self.lib.pcd_gemm_f16_out32(C.byref(g), out.data_ptr(), c, self._st()), "gemm_f16_out32")
        elif resid is None:
            self._chk(self.lib.pcd_gemm_f16(C.byref(g), out.data_ptr(), c, self._st()), "gemm_f16")
        else:
            self._chk(self.lib.pcd_gemm_f16_residual(C.byref(g), resid.data_ptr(), c, out.data_ptr(), c, self._st()), "gemm_resid")

    # ------------------------------------------------------------------ forward
    def _bn_relu(self, L: _Conv, z: torch.Tensor, m: int, train_stats: bool):
        lib, st = self.lib, self._st()
        c = L.cout
        L.z = z
        L.mean = self._buf(L.conv + ".mean", (c,), torch.float32)
        L.var = self._buf(L.conv + ".var", (c,), torch.float32)
        scratch = self._buf("bn.scratch", (2 * 4096,), torch.float32)
        rm = self.buf[L.bn + ".running_mean"] if train_stats else None
        rv = self.buf[L.bn + ".running_var"] if train_stats else None
        self._chk(lib.pcd_bn_batch_stats(z.data_ptr(), m, c, BN_MOMENTUM, L.mean.data_ptr(), L.var.data_ptr(),
                                         rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None,
                                         scratch.data_ptr(), st), "bn_stats")
        if train_stats:
            self.buf[L.bn + ".num_batches_tracked"] += 1
        L.a = self._buf(L.conv + ".a", (m, c), torch.float16)
        self._chk(lib.pcd_bn_apply_f16(z.data_ptr(), m, c, L.mean.data_ptr(), L.var.data_ptr(),
                                       self.p[L.bn + ".weight"].data_ptr(), self.p[L.bn + ".bias"].data_ptr(), BN_EPS, 1,
                                       L.a.data_ptr(), st), "bn_apply")
        return L.a

    def _conv(self, L: _Conv, inputs, m: int, bn: bool, update_stats: bool, shape_bias=None, rps=0, w=None, ldw=None):
        """z = [inputs] W^T + b, then BatchNorm(batch statistics) + ReLU if the layer has one."""
        L.inputs = inputs
        (a1, k1) = inputs[0]
        (a2, k2) = inputs[1] if len(inputs) > 1 else (None, 0)
        # conv outputs that feed a BatchNorm stay fp32 (x_hat = (z - mean) * rstd cancels); bare convs feed a GEMM: fp16
        z = self._buf(L.conv + ".z", (m, L.cout), torch.float32 if bn else torch.float16)
        wt = self.w16[L.conv]
        bias = None if shape_bias is not None else self.p[L.conv + ".bias"].data_ptr()
        self._gemm(a1, k1, a2, k2, wt.data_ptr() if w is None else w, wt.shape[1] if ldw is None else ldw, bias,
                   shape_bias, rps, m, L.cout, z)
        if not bn:
            L.z = L.a = z
            return z
        return self._bn_relu(L, z, m, update_stats)

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, update_stats: bool = True) -> torch.Tensor:
        """eps_hat (B, N, 3) fp32 with the network in train() mode; keeps what backward needs."""
        lib, st = self.lib, self._st()
        b, n, _ = x_t.shape
        m = b * n
        if m % 64 != 0:
            raise ValueError("B*N must be a multiple of 64 (reduction length of the backward-weight GEMM)")
        self.b, self.n, self.m = b, n, m
        self.x = x_t.to(torch.float32).contiguous()
        # time embedding: sinusoid on the host with the reference's ops (networks.py:820-838), MLP on the device
        tt = t.detach().to("cpu", torch.float32)
        e = tt[:, None] * self.freqs[None, :]
        self.emb = torch.cat((torch.sin(e), torch.cos(e)), dim=-1).to(self.dev)
        p = self.p
        self.h1 = self._buf("t.h1", (b, 256), torch.float32)
        self.s1 = self._buf("t.s1", (b, 256), torch.float32)
        self.temb = self._buf("t.temb", (b, 256), torch.float32)
        self._mm(self.emb.data_ptr(), 256, 0, p["time_mlp.0.weight"].data_ptr(), 256, 1, b, 256, 256,
                 p["time_mlp.0.bias"].data_ptr(), 0, self.h1.data_ptr(), 256)
        self._chk(lib.pcd_silu_f32(self.h1.data_ptr(), self.h1.numel(), self.s1.data_ptr(), st), "silu")
        self._mm(self.s1.data_ptr(), 256, 0, p["time_mlp.2.weight"].data_ptr(), 256, 1, b, 256, 256,
                 p["time_mlp.2.bias"].data_ptr(), 0, self.temb.data_ptr(), 256)
        # enc1.conv1: [xyz | temb] -> 64; the time columns give a per-shape bias
        e1 = self.enc[0][0]
        w1 = p["enc1.conv1.weight"]
        self.tbias = self._buf("t.tbias", (b, 64), torch.float32)
        self._mm(self.temb.data_ptr(), 256, 0, w1.data_ptr() + 3 * 4, 259, 1, b, 64, 256, p["enc1.conv1.bias"].data_ptr(), 0,
                 self.tbias.data_ptr(), 64)
        z0 = self._buf("enc1.conv1.z", (m, 64), torch.float32)
        self._chk(lib.pcd_enc1_linear(self.x.data_ptr(), m, n, self.w_xyz.data_ptr(), 64, self.tbias.data_ptr(), z0.data_ptr(), st),
                  "enc1_linear")
        a = self._bn_relu(e1, z0, m, update_stats)
        skips = []
        for bi, blk in enumerate(self.enc):
            for li, L in enumerate(blk):
                if bi == 0 and li == 0:
                    continue
                a = self._conv(L, [(a, L.cin)], m, True, update_stats)
            skips.append(a)
        x1, x2, x3, x4 = skips
        a = self._conv(self.gf[0], [(x4, 1024)], m, True, update_stats)
        a = self._conv(self.gf[1], [(a, 2048)], m, True, update_stats)
        # max over the N points of each shape, with the argmax for backward (networks.py:807)
        self.gmax = self._buf("g.max", (b, 4096), torch.float32)
        self.garg = self._buf("g.arg", (b, 4096), torch.int32)
        self._chk(lib.pcd_colmax_argmax_f16(a.data_ptr(), b, n, 4096, self.gmax.data_ptr(), self.garg.data_ptr(), st), "colmax")
        # dec4.conv1 = [global (4096, constant over N) | refine4(x4) (1024)]: the global half is a per-shape bias
        w4 = p["dec4.conv1.weight"]
        self.gbias = self._buf("g.bias", (b, 1024), torch.float32)
        self._mm(self.gmax.data_ptr(), 4096, 0, w4.data_ptr(), 5120, 1, b, 1024, 4096, p["dec4.conv1.bias"].data_ptr(), 0,
                 self.gbias.data_ptr(), 1024)
        prev = None
        for blk, xs in zip(self.dec, (x4, x3, x2, x1)):
            c = xs.shape[1]
            r = self._conv(self.refine[c], [(xs, c)], m, False, False)
            if prev is None:
                a = self._conv(blk[0], [(r, 1024)], m, True, update_stats, shape_bias=self.gbias.data_ptr(), rps=n)
            else:
                a = self._conv(blk[0], [(prev, prev.shape[1]), (r, c)], m, True, update_stats)
            a = self._conv(blk[1], [(a, blk[1].cin)], m, True, update_stats)
            prev = a = self._conv(blk[2], [(a, blk[2].cin)], m, True, update_stats)
        a = self._conv(self.out0, [(a, 64)], m, True, update_stats)
        self.pred = self._buf("pred", (b, n, 3), torch.float32)
        self.w_head = p["output.3.weight"].view(3, 64)
        self._chk(lib.pcd_head3(a.data_ptr(), m, 64, self.w_head.data_ptr(), p["output.3.bias"].data_ptr(), self.pred.data_ptr(), st),
                  "head3")
        return self.pred

    # ------------------------------------------------------------------ backward
    def _conv_backward(self, L: _Conv, dz: torch.Tensor, m: int, targets, w_cols0: int = 0):
        """Given dz (M, C): dW, db into the gradient views, and the input gradients.
        targets: per input either None (not needed), ('set', tensor) or ('add', tensor)."""
        lib, st = self.lib, self._st()
        c = L.cout
        gw = self.g[L.conv + ".weight"]
        ktot = gw.shape[1]
        if self.debug is not None:
            self.debug[L.conv + ".dz"] = dz.clone()
        if not L.bn:
            self._chk(lib.pcd_colsum_f16(dz.data_ptr(), m, 1, c, self.g[L.conv + ".bias"].data_ptr(), st), "colsum")
        # (a conv bias in front of a BatchNorm: dz is mean-free per channel by construction, its column sum is exactly the
        #  analytic zero - the gradient view keeps the 0 it was allocated with; autograd leaves ~1e-9 rounding noise there)
        dzT = self._buf("bwd.dzT", (4096 * m,), torch.float16)
        self._chk(lib.pcd_transpose_f16(dz.data_ptr(), m, c, dzT.data_ptr(), st), "transpose")
        aT = self._buf("bwd.aT", (4096 * m,), torch.float16)
        col = w_cols0
        wt = self.w16t[L.conv]
        row = 0
        for (a_in, k), tgt in zip(L.inputs, targets):
            # dW[:, col:col+k] = dz^T a_in : rows = C, reduction = M, columns = k
            self._chk(lib.pcd_transpose_f16(a_in.data_ptr(), m, k, aT.data_ptr(), st), "transpose")
            g = _gemm_desc(dzT.data_ptr(), m, m, None, 0, 0, aT.data_ptr(), m, None, None, 0, c, k)
            # a C x k output is a handful of tiles with an M-deep reduction: split the reduction so that one launch
            # carries >= ~512 tiles, then add the fp32 slabs in a fixed order
            tiles = -(-c // 128) * -(-k // 128)
            splits = 1
            while splits * tiles < 512 and (m // 64) % (splits * 2) == 0 and m // (splits * 2) >= 256:
                splits *= 2
            if splits == 1:
                self._chk(lib.pcd_gemm_f16_out32(C.byref(g), gw.data_ptr() + col * 4, ktot, st), "gemm_dW")
            else:
                slabs = self._buf("bwd.slabs", (16 * 1024 * 1024,), torch.float32)
                if splits * c * k > slabs.numel():
                    slabs = self._buf("bwd.slabs", (splits * c * k,), torch.float32)
                self._chk(lib.pcd_gemm_f16_splitk(C.byref(g), splits, slabs.data_ptr(), st), "gemm_dW_splitk")
                self._chk(lib.pcd_sum_slabs_f32(slabs.data_ptr(), splits, c, k, gw.data_ptr() + col * 4, ktot, st), "sum_slabs")
            if tgt is not None:
                mode, dst = tgt
                # da_in = dz W[:, col:col+k] : the rows [row, row+k) of W^T
                if self.debug is not None and mode == "add":
                    self.debug[f"{L.conv}.in{len(self.debug)}.before_add"] = dst.clone()
                self._gemm(dz, c, None, 0, wt.data_ptr() + row * c * 2, c, None, None, 0, m, k, dst,
                           resid=dst if mode == "add" else None)
                if self.debug is not None:
                    self.debug[f"{L.conv}.din{col - w_cols0}"] = dst.clone()
            col += k
            row += k

    def _bn_backward(self, L: _Conv, da: torch.Tensor, m: int) -> torch.Tensor:
        """da (grad of the post-ReLU activation) -> dz in place; dgamma, dbeta into the gradient views."""
        if self.debug is not None:
            self.debug[L.conv + ".da"] = da.clone()
        self._chk(self.lib.pcd_bn_backward_f16(da.data_ptr(), L.z.data_ptr(), m, L.cout, L.mean.data_ptr(), L.var.data_ptr(),
                                               self.p[L.bn + ".weight"].data_ptr(), self.p[L.bn + ".bias"].data_ptr(), BN_EPS, 1,
                                               self.g[L.bn + ".weight"].data_ptr(), self.g[L.bn + ".bias"].data_ptr(),
                                               da.data_ptr(), self._st()), "bn_backward")
        return da

    def backward(self, target: torch.Tensor) -> torch.Tensor:
        """L1 loss against `target` (the noise) and all parameter gradients (scaled by loss_scale) into self.G.
        Returns the loss as a 0-d device tensor."""
        lib, st = self.lib, self._st()
        b, n, m, p, g = self.b, self.n, self.m, self.p, self.g
        loss_sum = self._buf("loss", (1,), torch.float32)
        dpred = self._buf("dpred", (m, 3), torch.float32)
        target = target.to(torch.float32).contiguous()
        self._chk(lib.pcd_l1_loss(self.pred.data_ptr(), target.data_ptr(), m * 3, self.loss_scale, loss_sum.data_ptr(),
                                  dpred.data_ptr(), st), "l1_loss")
        # head 64 -> 3
        a_out = self.out0.a
        self._chk(lib.pcd_vec3_outer(a_out.data_ptr(), dpred.data_ptr(), m, 64, g["output.3.weight"].data_ptr(),
                                     g["output.3.bias"].data_ptr(), st), "vec3_outer")
        da = self._buf("bwd.da0", (m, 64), torch.float16)
        self._chk(lib.pcd_vec3_expand_f16(dpred.data_ptr(), self.w_head.data_ptr(), m, 64, da.data_ptr(), st), "vec3_expand")

        def chain(L: _Conv, da_out, targets):
            dz = self._bn_backward(L, da_out, m) if L.bn else da_out
            self._conv_backward(L, dz, m, targets)

        def fresh(key, c):
            return self._buf(key, (m, c), torch.float16)

        d_in = fresh("bwd.d_out0", 64)
        chain(self.out0, da, [("set", d_in)])
        da = d_in
        dskip: Dict[int, torch.Tensor] = {}
        # decoder, last block first
        for bi in (3, 2, 1, 0):
            blk = self.dec[bi]
            c_skip = (1024, 512, 256, 128)[bi]
            d2 = fresh(f"bwd.{blk[2].conv}", blk[2].cin)
            chain(blk[2], da, [("set", d2)])
            d1 = fresh(f"bwd.{blk[1].conv}", blk[1].cin)
            chain(blk[1], d2, [("set", d1)])
            dr = fresh(f"bwd.r{c_skip}", c_skip)
            if bi == 0:
                dz = self._bn_backward(blk[0], d1, m)
                # global half: per-shape sums of dz drive dW[:, :4096] and the max-pool gradient
                S = self._buf("bwd.S", (b, 1024), torch.float32)
                self._chk(lib.pcd_colsum_f16(dz.data_ptr(), n, b, 1024, S.data_ptr(), st), "colsum_shape")
                gw = g["dec4.conv1.weight"]
                self._mm(S.data_ptr(), 1024, 1, self.gmax.data_ptr(), 4096, 0, 1024, 4096, b, None, 0, gw.data_ptr(), 5120)
                dG = self._buf("bwd.dG", (b, 4096), torch.float32)
                self._mm(S.data_ptr(), 1024, 0, p["dec4.conv1.weight"].data_ptr(), 5120, 0, b, 4096, 1024, None, 0, dG.data_ptr(), 4096)
                self._conv_backward(blk[0], dz, m, [("set", dr)], w_cols0=4096)
                da = None
            else:
                dprev = fresh(f"bwd.prev{bi}", blk[0].inputs[0][1])
                chain(blk[0], d1, [("set", dprev), ("set", dr)])
                da = dprev
            # refine_k: bare conv on the skip tensor
            dx = fresh(f"bwd.x{c_skip}", c_skip)
            self._conv_backward(self.refine[c_skip], dr, m, [("set", dx)])
            dskip[c_skip] = dx
        # global_feat: scatter dG through the argmax, then two conv+BN+ReLU stages into dx4
        dgf = self._buf("bwd.dgf", (m, 4096), torch.float16)
        self._chk(lib.pcd_maxpool_backward_f16(dG.data_ptr(), self.garg.data_ptr(), b, n, 4096, dgf.data_ptr(), st), "maxpool_bwd")
        d_g0 = fresh("bwd.g0", 2048)
        chain(self.gf[1], dgf, [("set", d_g0)])
        chain(self.gf[0], d_g0, [("add", dskip[1024])])
        # encoder: each block ends in a skip tensor whose gradient is already seeded by the decoder side
        for bi in (3, 2, 1, 0):
            blk = self.enc[bi]
            c_out = (128, 256, 512, 1024)[bi]
            da = dskip[c_out]
            d2 = fresh(f"bwd.{blk[2].conv}", blk[2].cin)
            chain(blk[2], da, [("set", d2)])
            d1 = fresh(f"bwd.{blk[1].conv}", blk[1].cin)
            chain(blk[1], d2, [("set", d1)])
            if bi > 0:
                chain(blk[0], d1, [("add", dskip[blk[0].cin])])
        # enc1.conv1: K = 3 + 256, all fp32 side products
        e1 = self.enc[0][0]
        dz0 = self._bn_backward(e1, d1, m)
        gw = g["enc1.conv1.weight"]
        tmp = self._buf("bwd.wxyzT", (3, 64), torch.float32)
        self._chk(lib.pcd_vec3_outer(dz0.data_ptr(), self.x.data_ptr(), m, 64, tmp.data_ptr(), None, st), "vec3_outer")
        gw[:, :3, 0].copy_(tmp.t())
        dtb = self._buf("bwd.dtbias", (b, 64), torch.float32)
        self._chk(lib.pcd_colsum_f16(dz0.data_ptr(), n, b, 64, dtb.data_ptr(), st), "colsum_shape")
        ones = self._buf("ones", (1, b), torch.float32)
        ones.fill_(1.0)
        self._mm(dtb.data_ptr(), 64, 1, self.temb.data_ptr(), 256, 0, 64, 256, b, None, 0, gw.data_ptr() + 3 * 4, 259)
        self._mm(ones.data_ptr(), b, 0, dtb.data_ptr(), 64, 0, 1, 64, b, None, 0, g["enc1.conv1.bias"].data_ptr(), 64)
        dtemb = self._buf("bwd.dtemb", (b, 256), torch.float32)
        self._mm(dtb.data_ptr(), 64, 0, p["enc1.conv1.weight"].data_ptr() + 3 * 4, 259, 0, b, 256, 64, None, 0, dtemb.data_ptr(), 256)
        # time_mlp: Linear -> SiLU -> Linear
        self._mm(dtemb.data_ptr(), 256, 1, self.s1.data_ptr(), 256, 0, 256, 256, b, None, 0, g["time_mlp.2.weight"].data_ptr(), 256)
        self._mm(ones.data_ptr(), b, 0, dtemb.data_ptr(), 256, 0, 1, 256, b, None, 0, g["time_mlp.2.bias"].data_ptr(), 256)
        ds = self._buf("bwd.ds", (b, 256), torch.float32)
        self._mm(dtemb.data_ptr(), 256, 0, p["time_mlp.2.weight"].data_ptr(), 256, 0, b, 256, 256, None, 0, ds.data_ptr(), 256)
        dh = self._buf("bwd.dh", (b, 256), torch.float32)
        self._chk(lib.pcd_silu_backward_f32(self.h1.data_ptr(), ds.data_ptr(), ds.numel(), dh.data_ptr(), st), "silu_bwd")
        self._mm(dh.data_ptr(), 256, 1, self.emb.data_ptr(), 256, 0, 256, 256, b, None, 0, g["time_mlp.0.weight"].data_ptr(), 256)
        self._mm(ones.data_ptr(), b, 0, dh.data_ptr(), 256, 0, 1, 256, b, None, 0, g["time_mlp.0.bias"].data_ptr(), 256)
        return loss_sum[0] / float(m * 3)

    # ------------------------------------------------------------------ optimizer
    def grads(self) -> Dict[str, torch.Tensor]:
        """Unscaled parameter gradients (copies), keyed like `named_parameters()`."""
        return {k: v.clone() / self.loss_scale for k, v in self.g.items()}

    def optimizer_step(self):
        self.step_count += 1
        b1, b2 = self.betas
        world = _allreduce_gradients(self.G)           # data parallel: mean gradient over the ranks (BatchNorm stays per rank)
        self._chk(self.lib.pcd_adamw_step(self.P.data_ptr(), self.G.data_ptr(), self.M1.data_ptr(), self.M2.data_ptr(),
                                          self.P.numel(), self.lr, b1, b2, self.eps, self.wd, self.step_count,
                                          self.loss_scale * world, self._st()), "adamw")
        self.refresh_weights()

    def train_step(self, x_t: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        self.forward(x_t, t, update_stats=True)
        loss = self.backward(noise)
        self.optimizer_step()
        return loss


    # torch.optim-like aliases so the object can stand where the reference's optimizer does
    def step(self):
        self.optimizer_step()

    def zero_grad(self):
        pass                                # every backward overwrites the whole gradient buffer

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "exp_avg": self.M1, "exp_avg_sq": self.M2}


class LatentTrainer:
    """Forward + backward + AdamW for `SimpleLatentUNetPointNet` (networks.py:963-1086) in train() mode, as used by
    `LatentDiffusion.training_step` (diffusion.py:424-443; the VAE stays frozen).  The batch is 16-32 latent vectors,
    so every product is a few-row fp32 product (`pcd_matmul_f32`: weights read once) and everything stays fp32 -
    GroupNorm is per sample, so unlike the point denoiser nothing here couples the batch.  Dropout(0.1) after `dec1`
    (networks.py:1035) takes its keep mask from torch's generator, or from the caller (parity tests)."""

    GROUPS = 8
    DROPOUT = 0.1

    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.model = model
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.dev = model.device
        if self.dev.type != "cuda":
            raise RuntimeError("LatentTrainer needs the model on an MI355X (model.to('cuda'))")
        self.step_count = 0
        self.P, self.G, self.M1, self.M2, self.p, self.g = _flatten_parameters(model, self.dev)
        self.freqs = timestep_freqs(256)
        self._ws: Dict[str, torch.Tensor] = {}
        self.saved: Dict[str, tuple] = {}

    _st = PointTrainer._st
    _chk = PointTrainer._chk
    _buf = PointTrainer._buf
    _mm = PointTrainer._mm

    # ------------------------------------------------------------------ layer helpers
    def _lin(self, name: str, inputs) -> torch.Tensor:
        """y = [inputs] W^T + b  (nn.Linear; the torch.cat of networks.py:1062,1074-1077 is a split product)."""
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        c, ktot = w.shape
        rows = inputs[0][0].shape[0]
        y = self._buf(name + ".y", (rows, c), torch.float32)
        off = 0
        for i, (a, k) in enumerate(inputs):
            self._mm(a.data_ptr(), k, 0, w.data_ptr() + off * 4, ktot, 1, rows, c, k, b.data_ptr() if i == 0 else None, 1 if i else 0,
                     y.data_ptr(), c)
            off += k
        self.saved[name] = tuple(inputs)
        return y

    def _lin_backward(self, name: str, dy: torch.Tensor, targets) -> None:
        """dW, db into the gradient views; input gradients into targets[i] = None | ('set'|'add', tensor)."""
        w, gw, gb = self.p[name + ".weight"], self.g[name + ".weight"], self.g[name + ".bias"]
        c, ktot = w.shape
        rows = dy.shape[0]
        ones = self._buf("ones", (1, rows), torch.float32)
        ones.fill_(1.0)
        self._mm(ones.data_ptr(), rows, 0, dy.data_ptr(), c, 0, 1, c, rows, None, 0, gb.data_ptr(), c)
        off = 0
        for (a, k), tgt in zip(self.saved[name], targets):
            self._mm(dy.data_ptr(), c, 1, a.data_ptr(), k, 0, c, k, rows, None, 0, gw.data_ptr() + off * 4, ktot)      # dW = dy^T a
            if tgt is not None:
                mode, dst = tgt
                self._mm(dy.data_ptr(), c, 0, w.data_ptr() + off * 4, ktot, 0, rows, k, c, None, 1 if mode == "add" else 0,
                         dst.data_ptr(), k)                                                                          # da = dy W
            off += k

    def _gn(self, name: str, x: torch.Tensor) -> torch.Tensor:
        rows, c = x.shape
        y = self._buf(name + ".y", (rows, c), torch.float32)
        mean = self._buf(name + ".mean", (rows, self.GROUPS), torch.float32)
        rstd = self._buf(name + ".rstd", (rows, self.GROUPS), torch.float32)
        self._chk(self.lib.pcd_groupnorm_f32(x.data_ptr(), rows, c, self.GROUPS, self.p[name + ".weight"].data_ptr(),
                                             self.p[name + ".bias"].data_ptr(), 1e-5, 1, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             self._st()), "groupnorm")
        self.saved[name] = (x, mean, rstd)
        return y

    def _gn_backward(self, name: str, dy: torch.Tensor) -> torch.Tensor:
        x, mean, rstd = self.saved[name]
        rows, c = x.shape
        dx = self._buf(name + ".dx", (rows, c), torch.float32)
        self._chk(self.lib.pcd_groupnorm_backward_f32(dy.data_ptr(), x.data_ptr(), rows, c, self.GROUPS, self.p[name + ".weight"].data_ptr(),
                                                      self.p[name + ".bias"].data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, dx.data_ptr(),
                                                      self.g[name + ".weight"].data_ptr(), self.g[name + ".bias"].data_ptr(), self._st()),
                  "groupnorm_backward")
        return dx

    # ------------------------------------------------------------------ forward / backward
    def forward(self, z_t: torch.Tensor, t: torch.Tensor, dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        lib, st = self.lib, self._st()
        b = z_t.shape[0]
        self.z = z_t.to(torch.float32).contiguous()
        tt = t.detach().to("cpu", torch.float32)
        e = tt[:, None] * self.freqs[None, :]
        self.emb = torch.cat((torch.sin(e), torch.cos(e)), dim=-1).to(self.dev)
        h1 = self._lin("time_mlp.0", [(self.emb, 256)])
        s1 = self._buf("t.s1", (b, 256), torch.float32)
        self._chk(lib.pcd_silu_f32(h1.data_ptr(), h1.numel(), s1.data_ptr(), st), "silu")
        te = self._lin("time_mlp.2", [(s1, 256)])
        z1 = self._gn("enc1.1", self._lin("enc1.0", [(self.z, 256), (te, 256)]))
        z2 = self._gn("enc2.1", self._lin("enc2.0", [(z1, 128)]))
        z3 = self._gn("enc3.1", self._lin("enc3.0", [(z2, 256)]))
        z4 = self._gn("enc4.1", self._lin("enc4.0", [(z3, 512)]))
        g = self._gn("global_feat.1", self._lin("global_feat.0", [(z4, 1024)]))
        g = self._gn("global_feat.4", self._lin("global_feat.3", [(g, 2048)]))
        h = self._gn("dec4.1", self._lin("dec4.0", [(g, 4096), (self._lin("refine4", [(z4, 1024)]), 1024)]))
        h = self._gn("dec3.1", self._lin("dec3.0", [(h, 1024), (self._lin("refine3", [(z3, 512)]), 512)]))
        h = self._gn("dec2.1", self._lin("dec2.0", [(h, 512), (self._lin("refine2", [(z2, 256)]), 256)]))
        h = self._gn("dec1.1", self._lin("dec1.0", [(h, 256), (self._lin("refine1", [(z1, 128)]), 128)]))
        if dropout_mask is None:
            dropout_mask = (torch.rand(b, 128, device=self.dev) >= self.DROPOUT).to(torch.float32)
        self.mask = dropout_mask.to(self.dev, torch.float32).contiguous()
        hd = self._buf("drop.y", (b, 128), torch.float32)
        self._chk(lib.pcd_mask_scale_f32(h.data_ptr(), self.mask.data_ptr(), 1.0 / (1.0 - self.DROPOUT), h.numel(), hd.data_ptr(), st), "dropout")
        self.o0 = self._lin("output.0", [(hd, 128)])
        o0r = self._buf("o0.relu", (b, 128), torch.float32)
        self._chk(lib.pcd_relu_f32(self.o0.data_ptr(), self.o0.numel(), o0r.data_ptr(), st), "relu")
        self.pred = self._lin("output.2", [(o0r, 128)])
        return self.pred

    def backward(self, target: torch.Tensor) -> torch.Tensor:
        lib, st = self.lib, self._st()
        b = self.pred.shape[0]
        n = self.pred.numel()
        loss_sum = self._buf("loss", (1,), torch.float32)
        d = self._buf("d.pred", (b, 256), torch.float32)
        target = target.to(self.dev, torch.float32).contiguous()
        self._chk(lib.pcd_l1_loss(self.pred.data_ptr(), target.data_ptr(), n, 1.0, loss_sum.data_ptr(), d.data_ptr(), st), "l1_loss")

        def new(key, c):
            return self._buf("d." + key, (b, c), torch.float32)

        d_o0r = new("o0r", 128)
        self._lin_backward("output.2", d, [("set", d_o0r)])
        d_o0 = new("o0", 128)
        self._chk(lib.pcd_relu_backward_f32(self.o0.data_ptr(), d_o0r.data_ptr(), d_o0.numel(), d_o0.data_ptr(), st), "relu_bwd")
        d_hd = new("hd", 128)
        self._lin_backward("output.0", d_o0, [("set", d_hd)])
        d_h = new("h1", 128)
        self._chk(lib.pcd_mask_scale_f32(d_hd.data_ptr(), self.mask.data_ptr(), 1.0 / (1.0 - self.DROPOUT), d_hd.numel(), d_h.data_ptr(), st), "dropout_bwd")
        dz = {}
        for name, kprev, kskip, refine in (("dec1", 256, 128, "refine1"), ("dec2", 512, 256, "refine2"),
                                           ("dec3", 1024, 512, "refine3"), ("dec4", 4096, 1024, "refine4")):
            dx = self._gn_backward(name + ".1", d_h)
            d_prev, d_r = new(name + ".prev", kprev), new(name + ".r", kskip)
            self._lin_backward(name + ".0", dx, [("set", d_prev), ("set", d_r)])
            dz[kskip] = new(f"z{kskip}", kskip)
            self._lin_backward(refine, d_r, [("set", dz[kskip])])
            d_h = d_prev
        dx = self._gn_backward("global_feat.4", d_h)
        d_g0 = new("g0", 2048)
        self._lin_backward("global_feat.3", dx, [("set", d_g0)])
        dx = self._gn_backward("global_feat.1", d_g0)
        self._lin_backward("global_feat.0", dx, [("add", dz[1024])])
        for name, kin, kout in (("enc4", 512, 1024), ("enc3", 256, 512), ("enc2", 128, 256)):
            dx = self._gn_backward(name + ".1", dz[kout])
            self._lin_backward(name + ".0", dx, [("add", dz[kin])])
        dx = self._gn_backward("enc1.1", dz[128])
        d_te = new("te", 256)
        self._lin_backward("enc1.0", dx, [None, ("set", d_te)])
        d_s1 = new("s1", 256)
        self._lin_backward("time_mlp.2", d_te, [("set", d_s1)])
        d_h1 = new("th1", 256)
        h1 = self._ws["time_mlp.0.y"]
        self._chk(lib.pcd_silu_backward_f32(h1.data_ptr(), d_s1.data_ptr(), d_s1.numel(), d_h1.data_ptr(), st), "silu_bwd")
        self._lin_backward("time_mlp.0", d_h1, [None])
        return loss_sum[0] / float(n)

    def grads(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self.g.items()}

    def optimizer_step(self):
        self.step_count += 1
        b1, b2 = self.betas
        world = _allreduce_gradients(self.G)
        self._chk(self.lib.pcd_adamw_step(self.P.data_ptr(), self.G.data_ptr(), self.M1.data_ptr(), self.M2.data_ptr(), self.P.numel(),
                                          self.lr, b1, b2, self.eps, self.wd, self.step_count, float(world), self._st()), "adamw")
        self.model.invalidate()        # the sampler's packed fp16 weights are stale now

    step = optimizer_step

    def zero_grad(self):
        pass

    def train_step(self, z_t, t, noise, dropout_mask=None) -> torch.Tensor:
        self.forward(z_t, t, dropout_mask)
        loss = self.backward(noise)
        self.optimizer_step()
        return loss


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(T_max, eta_min) in closed form (diffusion.py:415-419)."""

    def __init__(self, trainer, T_max: int, eta_min: float = 1e-6):
        self.trainer, self.T_max, self.eta_min, self.base, self.epoch = trainer, T_max, eta_min, trainer.lr, 0

    def step(self, metric=None) -> None:
        import math
        self.epoch += 1
        self.trainer.lr = self.eta_min + (self.base - self.eta_min) * (1 + math.cos(math.pi * self.epoch / self.T_max)) / 2


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience) as configured at diffusion.py:61,
    restated for the trainer (threshold 1e-4 relative, cooldown 0, min_lr 0: torch's defaults)."""

    def __init__(self, trainer: PointTrainer, factor: float = 0.5, patience: int = 5, threshold: float = 1e-4):
        self.trainer, self.factor, self.patience, self.threshold = trainer, factor, patience, threshold
        self.best, self.bad = float("inf"), 0

    def step(self, metric: float) -> None:
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.trainer.lr *= self.factor
            self.bad = 0


def save_checkpoint(model, path: str, epoch: int, extra: Optional[dict] = None) -> None:
    """A `.ckpt` in the layout the reference's Lightning checkpoints have (`state_dict` + `hyper_parameters`),
    which `PointCloudDiffusion.load_from_checkpoint` of either code base reads back."""
    import os
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    payload = {"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
               "hyper_parameters": dict(model.hparams), "epoch": epoch,
               "pytorch-lightning_version": "2.3.3"}       # the reference's pin; Lightning's loader looks for this key
    payload.update(extra or {})
    torch.save(payload, path)


class _RankRng:
    """Context manager: inside, torch's CPU / device generators (and `torch.initial_seed()`, which seeds the on-device Philox
    stream) are this rank's private stream; outside, the process keeps the RNG state all ranks share."""

    def __init__(self, device, rank: int):
        self.device = torch.device(device)
        shared = self._get()
        torch.manual_seed((torch.initial_seed() + 0x9E3779B1 * (rank + 1)) & 0x7FFFFFFFFFFFFFFF)
        self.mine = self._get()
        self._set(shared)

    def _get(self):
        return torch.get_rng_state(), (torch.cuda.get_rng_state(self.device) if self.device.type == "cuda" else None)

    def _set(self, st):
        torch.set_rng_state(st[0])
        if st[1] is not None:
            torch.cuda.set_rng_state(st[1], self.device)

    def __enter__(self):
        self.shared = self._get()
        self._set(self.mine)

    def __exit__(self, *exc):
        self.mine = self._get()
        self._set(self.shared)
        return False


def fit(model, data_module, max_epochs: int = 500, ckpt_dir: Optional[str] = None, log=print, max_steps: Optional[int] = None,
        save_top_k: int = 10, ckpt_name: str = "point_cloud_diffusion"):
    """What `pl.Trainer(max_epochs=...).fit(model, data_module)` does for the reference's train_point_ddpm.py:78-87 and
    train_point_ldm.py:100-108: epochs of training_step + optimizer step, then validation_step over the validation
    loader in eval() mode, the model's scheduler (plateau on `val_loss` / cosine per epoch), and the `save_top_k` best
    checkpoints by val_loss (train_point_ddpm.py:60-66)."""
    import inspect
    if "max_epochs" in inspect.signature(model.configure_optimizers).parameters:
        cfg = model.configure_optimizers(max_epochs=max_epochs)
    else:
        cfg = model.configure_optimizers()
    opt = cfg["optimizer"]
    sched = cfg["lr_scheduler"]["scheduler"] if isinstance(cfg["lr_scheduler"], dict) else cfg["lr_scheduler"]
    data_module.setup()
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    if world > 1:                              # every rank starts from rank 0's parameters and BatchNorm buffers
        for t in list(model.parameters()) + list(model.buffers()):
            _collective_inplace(dist.broadcast, t.data, 0)
        if hasattr(opt, "refresh_weights"):
            opt.refresh_weights()
        for sub in model.modules():            # packed fp16 weight images (incl. a frozen VAE's handle) were built from the
            inv = getattr(sub, "invalidate", None)      # pre-broadcast values on ranks > 0: rebuild them on next use
            if callable(inv):
                inv()
    # Ranks hold different data batches and must not draw the same (t, noise, dropout) for them, but they MUST enumerate the same
    # shuffled batch sequence (`group[rank]` below deals consecutive batches of ONE permutation out to the ranks; the loaders'
    # RandomSampler seeds itself from the global torch RNG).  So the global RNG stays shared, and only the model's own draws --
    # torch.rand for t, the on-device Philox stream seeded by torch.initial_seed() -- run under a per-rank RNG state that is
    # swapped in around training_step / validation_step and swapped out again (it does not leak out of fit() either).
    if world > 1:
        # ... which only holds if every rank ENTERS with the same global RNG state: nothing upstream enforces that (a caller that seeded per rank,
        # or not at all, would make each rank shuffle differently -- an epoch would no longer partition the dataset, silently).  Rank 0's seed wins.
        seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=model.device if dist.get_backend() == "nccl" else "cpu")
        _collective_inplace(dist.broadcast, seed, 0)
        torch.manual_seed(int(seed.item()))
    rank_rng = _RankRng(model.device, rank) if world > 1 else contextlib.nullcontext()
    kept: List[Tuple[float, str]] = []
    steps = 0
    history = []
    for epoch in range(max_epochs):
        if hasattr(model, "current_epoch"):
            model.current_epoch, model._max_epochs = epoch, max_epochs      # VAE3DLarge.get_kl_weight reads these
        model.train()
        tl = []
        group = []                             # data parallel: consecutive usable batches are dealt out `world` at a time
        for i, batch in enumerate(data_module.train_dataloader()):
            if batch.dim() == 3 and batch.shape[0] * batch.shape[1] % 64:
                continue                       # ragged last point-cloud batch: the backward-weight GEMM reduces over B*N in 64s
            group.append((i, batch))
            if len(group) < world:
                continue                       # an incomplete last group is dropped: every rank takes the same number of
            i, batch = group[rank]             # steps, so each optimizer all-reduce pairs the same step on all ranks
            group = []
            with rank_rng:
                loss = model.training_step(batch, i)
            opt.step()
            tl.append(loss)
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        model.eval()
        vl = []
        for i, b in enumerate(data_module.val_dataloader()):
            with rank_rng:
                vl.append(model.validation_step(b, i))
        train_loss = float(torch.stack(tl).mean()) if tl else float("nan")
        val_loss = float(torch.stack(vl).mean()) if vl else train_loss
        if world > 1:                          # one val_loss for the scheduler and the top-k logic on every rank
            v = torch.tensor([val_loss], dtype=torch.float64, device=model.device)
            _collective_inplace(dist.all_reduce, v)
            val_loss = float(v.item()) / world
        sched.step(val_loss)
        history.append((epoch, train_loss, val_loss, opt.lr))
        log(f"epoch {epoch}: train_loss {train_loss:.4f} val_loss {val_loss:.4f} lr {opt.lr:.2e}")
        if ckpt_dir is not None and rank == 0:
            import os
            path = os.path.join(ckpt_dir, f"{ckpt_name}-epoch={epoch:02d}-val_loss={val_loss:.2f}.ckpt")
            if len(kept) < save_top_k or val_loss < max(kept)[0]:
                save_checkpoint(model, path, epoch)
                kept.append((val_loss, path))
                kept.sort()
                for _, old in kept[save_top_k:]:
                    if os.path.exists(old) and old != path:
                        os.remove(old)
                kept = kept[:save_top_k]
        if max_steps is not None and steps >= max_steps:
            break
    return history
