"""Training step of the voxel VAE `VAE3DLarge` on the HIP kernels (reference networks.py:2209-2416: `calculate_loss`
= BCE(reconstruction) + kl_weight * KL with the module in train() mode, `torch.optim.Adam(lr)`).

Every Conv3d is run as "gather rows, then the GEMM": `pcd_im2col_f16` builds the row matrix col
[B*Do*Ho*Wo][k^3*Cin], the fp16 MFMA GEMM does forward (z = col W^T), backward-weight (dW = dz^T col, split-K over the
rows) and backward-data (dcol = dz W), and `pcd_col2im_f16` - the exact adjoint of the gather, written as a gather -
turns dcol into the input gradient.  A ConvTranspose3d(k, s, p) is the adjoint of a Conv3d(k, s, p) from its output
grid to its input grid, so it runs the same three pieces in the other order (product P = x Wg^T, then col2im; backward:
im2col of dz, then the two products) and carries no structural zeros.  BatchNorm3d (inside the residual blocks only) is the same channels-last
batch-statistics kernel pair as the point denoiser's BatchNorm1d.  The 512-wide bottleneck (fc_mu / fc_logvar /
decoder_input, reparameterisation, KL) is a handful of few-row fp32 products.

This is the straightforward formulation, not a tuned one: the row matrices are materialised (tens of GB of traffic per
step at batch 16).  It exists so that `train_point_ldm.py` can
run end to end on an MI355X; the sampler's implicit-GEMM convolutions (`csrc/conv3d.hip`) are the fast path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .specs import VAE_DEC, VAE_ENC
from .training import BN_EPS, BN_MOMENTUM, PointTrainer, _allreduce_gradients, _flatten_parameters


def _up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


class _VConv:
    """One Conv3d / ConvTranspose3d [+ BatchNorm3d] [+ ReLU]."""

    def __init__(self, key, transposed, cin, cout, k, s, p, din, dout, bn=None, relu=True):
        self.key, self.transposed, self.cin, self.cout, self.k, self.s, self.p = key, int(transposed), cin, cout, k, s, p
        self.din, self.dout, self.bn, self.relu = din, dout, bn, relu
        self.kk = k ** 3 * cin
        self.kp = _up(self.kk, 64)
        self.cp = _up(cout, 64)          # the output width is the reduction length of backward-data: multiple of 64
        if transposed:                   # run as the adjoint of Conv3d(k, s, p): product first, then col2im (no structural zeros)
            self.np_ = _up(k ** 3 * cout, 64)
            self.cp = cout
        self.col = self.z = self.a = self.mean = self.var = None


class VAETrainer:
    """Forward + backward + Adam for `VAE3DLarge`; parameters are the module's own, re-pointed into one flat buffer."""

    def __init__(self, vae, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, loss_scale: float = 32768.0):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.vae = vae
        self.lr, self.betas, self.eps, self.loss_scale = lr, betas, eps, float(loss_scale)
        self.dev = vae.device
        if self.dev.type != "cuda":
            raise RuntimeError("VAETrainer needs the model on an MI355X (vae.to('cuda'))")
        self.step_count = 0
        self.P, self.G, self.M1, self.M2, self.p, self.g = _flatten_parameters(vae, self.dev)
        self.buf = dict(vae.named_buffers())
        self._ws: Dict[str, torch.Tensor] = {}
        self.enc = self._program("encoder", VAE_ENC, 32)
        self.dec = self._program("decoder", VAE_DEC, 4)
        self.wm: Dict[str, torch.Tensor] = {}
        self.wmt: Dict[str, torch.Tensor] = {}
        self.bias: Dict[str, torch.Tensor] = {}
        self.refresh_weights()
        vae.register_load_state_dict_post_hook(lambda module, incompatible: self.refresh_weights())

    _st = PointTrainer._st
    _chk = PointTrainer._chk
    _mm = PointTrainer._mm

    def _buf(self, key, shape, dtype, zero=False):
        t = self._ws.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    def _scratch(self, key: str, numel: int, dtype=torch.float16) -> torch.Tensor:
        """A flat scratch buffer that only ever grows (shared by the layers: transposes, dcol, P, slabs)."""
        t = self._ws.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(numel, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    # ------------------------------------------------------------------ structure
    def _program(self, prefix: str, prog, d0: int):
        ops, d = [], d0
        for i, (idx, op, a) in enumerate(prog):
            key = f"{prefix}.{idx}"
            last = prefix == "decoder" and i == len(prog) - 1
            if op == "res":
                cin, cout = a
                c1 = _VConv(key + ".conv1", 0, cin, cout, 3, 1, 1, d, d, bn=key + ".bn1", relu=True)
                c2 = _VConv(key + ".conv2", 0, cout, cout, 3, 1, 1, d, d, bn=key + ".bn2", relu=False)
                dn = _VConv(key + ".downsample", 0, cin, cout, 1, 1, 0, d, d, relu=False) if cin != cout else None
                ops.append(("res", c1, c2, dn))
            else:
                cin, cout, k, s, p = a
                dout = (d + 2 * p - k) // s + 1 if op == "conv" else (d - 1) * s - 2 * p + k
                ops.append(("conv", _VConv(key, op == "convT", cin, cout, k, s, p, d, dout, relu=not last)))
                d = dout
        return ops

    def _layers(self):
        for ops in (self.enc, self.dec):
            for op in ops:
                for L in op[1:]:
                    if L is not None:
                        yield L

    def refresh_weights(self):
        """fp16 GEMM operands of the fp32 master weights: Wm [Cp][Kp] with column t*Cin + ci = tap t, input channel ci
        (Conv3d weight (Cout,Cin,k,k,k); ConvTranspose3d weight (Cin,Cout,k,k,k)), its transpose, padded bias / affine."""
        for L in self._layers():
            w = self.p[L.key + ".weight"]
            if L.transposed:
                # Wg [k^3*Cout (padded)][Cin], row t*Cout + co = W[:, co, t]: P = x Wg^T is what col2im scatters into the output
                full = self._buf(L.key + ".wg32", (L.np_, L.cin), torch.float32, zero=True)
                full[:L.k ** 3 * L.cout] = w.permute(2, 3, 4, 1, 0).reshape(L.k ** 3 * L.cout, L.cin)
                self.wm[L.key] = full.to(torch.float16)
                self.wmt[L.key] = full.t().contiguous().to(torch.float16)
                self.bias[L.key] = self.p[L.key + ".bias"]
                continue
            wm = (w.permute(1, 2, 3, 4, 0) if L.transposed else w.permute(0, 2, 3, 4, 1)).reshape(L.cout, L.kk)
            full = self._buf(L.key + ".wm32", (L.cp, L.kp), torch.float32, zero=True)
            full[:L.cout, :L.kk] = wm
            self.wm[L.key] = full.to(torch.float16)
            self.wmt[L.key] = full.t().contiguous().to(torch.float16)
            b = self._buf(L.key + ".bias_p", (L.cp,), torch.float32, zero=True)
            b[:L.cout] = self.p[L.key + ".bias"]
            self.bias[L.key] = b
            if L.bn:
                ga = self._buf(L.bn + ".gamma_p", (L.cp,), torch.float32, zero=True)
                be = self._buf(L.bn + ".beta_p", (L.cp,), torch.float32, zero=True)
                ga.fill_(1.0)
                ga[:L.cout] = self.p[L.bn + ".weight"]
                be[:L.cout] = self.p[L.bn + ".bias"]
        self.vae.invalidate()

    # ------------------------------------------------------------------ one layer
    def _convT_fwd(self, L: _VConv, a_in: torch.Tensor, b: int) -> torch.Tensor:
        """ConvTranspose3d(k, s, p) = adjoint of Conv3d(k, s, p) from the output grid to the input grid:
        P [M_in][k^3*Cout] = x Wg^T, y = col2im(P) + bias, ReLU."""
        lib, st = self.lib, self._st()
        m_in, m = b * L.din ** 3, b * L.dout ** 3
        L.m, L.mp, L.m_in, L.mp_in, L.a_in = m, _up(m, 64), m_in, _up(m_in, 64), a_in
        P = self._scratch("convT.P", L.mp_in * L.np_)
        g = _lib.GemmDesc()
        g.a1, g.lda1, g.k1 = a_in.data_ptr(), L.cin, L.cin
        g.w, g.ldw = self.wm[L.key].data_ptr(), L.cin
        g.m, g.c = m_in, L.np_
        self._chk(lib.pcd_gemm_f16(C.byref(g), P.data_ptr(), L.np_, st), "gemm_convT")
        y = self._buf(L.key + ".y", (L.mp, L.cout), torch.float16, zero=True)
        self._chk(lib.pcd_col2im_f16(P.data_ptr(), b, L.cout, L.dout, L.dout, L.dout, L.din, L.din, L.din, L.k, L.s, L.p, 0, L.np_, y.data_ptr(), st),
                  "col2im_convT")
        L.a = self._buf(L.key + ".a", (L.mp, L.cout), torch.float16, zero=True)
        self._chk(lib.pcd_bias_act_f16(y.data_ptr(), self.bias[L.key].data_ptr(), m, L.cout, int(L.relu), L.a.data_ptr(), st), "bias_act")
        return L.a

    def _convT_bwd(self, L: _VConv, d_out: torch.Tensor, b: int) -> torch.Tensor:
        lib, st = self.lib, self._st()
        dz = self._buf(L.key + ".dz", (L.mp, L.cout), torch.float16, zero=True)
        self._chk(lib.pcd_relu_mask_f16(d_out.data_ptr(), L.a.data_ptr(), L.m * L.cout, dz.data_ptr(), st), "relu_mask")
        db = self._buf("bwd.db", (512,), torch.float32)
        self._chk(lib.pcd_colsum_f16(dz.data_ptr(), L.m, 1, L.cout, db.data_ptr(), st), "colsum")
        self.g[L.key + ".bias"].copy_(db[:L.cout])
        n = L.mp_in * L.np_
        dP = self._buf(L.key + ".dP", (L.mp_in, L.np_), torch.float16, zero=True)          # rows >= M_in stay zero
        self._chk(lib.pcd_im2col_f16(dz.data_ptr(), b, L.cout, L.dout, L.dout, L.dout, L.din, L.din, L.din, L.k, L.s, L.p, 0, L.np_,
                                     dP.data_ptr(), st), "im2col_convT")
        # dWg [k^3*Cout][Cin] = dP^T x
        dPT = self._scratch("bwd.dzT", n)
        nx = L.cin * L.mp_in
        xT = self._scratch("bwd.colT", nx)
        self._chk(lib.pcd_transpose_f16(dP.data_ptr(), L.mp_in, L.np_, dPT.data_ptr(), st), "transpose")
        self._chk(lib.pcd_transpose_f16(L.a_in.data_ptr(), L.mp_in, L.cin, xT.data_ptr(), st), "transpose")
        dwg = self._buf(L.key + ".dwg", (L.np_, L.cin), torch.float32)
        g = _lib.GemmDesc()
        g.a1, g.lda1, g.k1 = dPT.data_ptr(), L.mp_in, L.mp_in
        g.w, g.ldw = xT.data_ptr(), L.mp_in
        g.m, g.c = L.np_, L.cin
        tiles = -(-L.np_ // 128) * -(-L.cin // 128)
        splits = 1
        while splits * tiles < 512 and (L.mp_in // 64) % (splits * 2) == 0 and L.mp_in // (splits * 2) >= 256:
            splits *= 2
        if splits == 1:
            self._chk(lib.pcd_gemm_f16_out32(C.byref(g), dwg.data_ptr(), L.cin, st), "gemm_dWg")
        else:
            slabs = self._scratch("bwd.slabs", splits * L.np_ * L.cin, torch.float32)
            self._chk(lib.pcd_gemm_f16_splitk(C.byref(g), splits, slabs.data_ptr(), st), "gemm_dWg_splitk")
            self._chk(lib.pcd_sum_slabs_f32(slabs.data_ptr(), splits, L.np_, L.cin, dwg.data_ptr(), L.cin, st), "sum_slabs")
        self.g[L.key + ".weight"].copy_(dwg[:L.k ** 3 * L.cout].reshape(L.k, L.k, L.k, L.cout, L.cin).permute(4, 3, 0, 1, 2))
        # dx = dP Wg
        dx = self._buf(L.key + ".dx", (L.mp_in, L.cin), torch.float16, zero=True)
        g2 = _lib.GemmDesc()
        g2.a1, g2.lda1, g2.k1 = dP.data_ptr(), L.np_, L.np_
        g2.w, g2.ldw = self.wmt[L.key].data_ptr(), L.np_
        g2.m, g2.c = L.m_in, L.cin
        self._chk(lib.pcd_gemm_f16(C.byref(g2), dx.data_ptr(), L.cin, st), "gemm_dx_convT")
        return dx

    def _conv_fwd(self, L: _VConv, a_in: torch.Tensor, b: int, update_stats: bool) -> torch.Tensor:
        if L.transposed:
            return self._convT_fwd(L, a_in, b)
        lib, st = self.lib, self._st()
        m = b * L.dout ** 3
        mp = _up(m, 64)
        L.m, L.mp, L.a_in = m, mp, a_in
        L.col = self._buf(L.key + ".col", (mp, L.kp), torch.float16, zero=True)
        self._chk(lib.pcd_im2col_f16(a_in.data_ptr(), b, L.cin, L.din, L.din, L.din, L.dout, L.dout, L.dout, L.k, L.s, L.p, L.transposed,
                                     L.kp, L.col.data_ptr(), st), "im2col")
        L.z = self._buf(L.key + ".z", (mp, L.cp), torch.float32 if L.bn else torch.float16, zero=True)
        g = _lib.GemmDesc()
        g.a1, g.lda1, g.k1 = L.col.data_ptr(), L.kp, L.kp
        g.w, g.ldw, g.bias = self.wm[L.key].data_ptr(), L.kp, self.bias[L.key].data_ptr()
        g.relu, g.m, g.c = int(L.relu and not L.bn), m, L.cp
        if L.bn:
            self._chk(lib.pcd_gemm_f16_out32(C.byref(g), L.z.data_ptr(), L.cp, st), "gemm_out32")
            L.mean = self._buf(L.key + ".mean", (L.cp,), torch.float32)
            L.var = self._buf(L.key + ".var", (L.cp,), torch.float32)
            scratch = self._buf("bn.scratch", (2 * 512,), torch.float32)
            rm = rv = None
            if update_stats:
                # running statistics live in the module's (unpadded) buffers: padded copies for the kernel
                rm = self._buf(L.bn + ".rm_p", (L.cp,), torch.float32, zero=True)
                rv = self._buf(L.bn + ".rv_p", (L.cp,), torch.float32, zero=True)
                rm[:L.cout] = self.buf[L.bn + ".running_mean"]
                rv[:L.cout] = self.buf[L.bn + ".running_var"]
            self._chk(lib.pcd_bn_batch_stats(L.z.data_ptr(), m, L.cp, BN_MOMENTUM, L.mean.data_ptr(), L.var.data_ptr(),
                                             rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None,
                                             scratch.data_ptr(), st), "bn_stats")
            if update_stats:
                self.buf[L.bn + ".running_mean"].copy_(rm[:L.cout])
                self.buf[L.bn + ".running_var"].copy_(rv[:L.cout])
                self.buf[L.bn + ".num_batches_tracked"] += 1
            L.a = self._buf(L.key + ".a", (mp, L.cp), torch.float16, zero=True)
            self._chk(lib.pcd_bn_apply_f16(L.z.data_ptr(), m, L.cp, L.mean.data_ptr(), L.var.data_ptr(),
                                           self._ws[L.bn + ".gamma_p"].data_ptr(), self._ws[L.bn + ".beta_p"].data_ptr(), BN_EPS,
                                           int(L.relu), L.a.data_ptr(), st), "bn_apply")
        else:
            self._chk(lib.pcd_gemm_f16(C.byref(g), L.z.data_ptr(), L.cp, st), "gemm_f16")
            L.a = L.z
        return L.a

    def _conv_bwd(self, L: _VConv, d_out: torch.Tensor, b: int, need_dx: bool = True) -> Optional[torch.Tensor]:
        """d_out: gradient of the layer's output activation, fp16 [Mp][Cp] (rows >= M zero).  Returns dx [M_in][Cin]."""
        if L.transposed:
            return self._convT_bwd(L, d_out, b)
        lib, st = self.lib, self._st()
        m, mp = L.m, L.mp
        if L.bn:
            dga = self._buf(L.bn + ".dgamma_p", (L.cp,), torch.float32)
            dbe = self._buf(L.bn + ".dbeta_p", (L.cp,), torch.float32)
            self._chk(lib.pcd_bn_backward_f16(d_out.data_ptr(), L.z.data_ptr(), m, L.cp, L.mean.data_ptr(), L.var.data_ptr(),
                                              self._ws[L.bn + ".gamma_p"].data_ptr(), self._ws[L.bn + ".beta_p"].data_ptr(), BN_EPS,
                                              int(L.relu), dga.data_ptr(), dbe.data_ptr(), d_out.data_ptr(), st), "bn_backward")
            self.g[L.bn + ".weight"].copy_(dga[:L.cout])
            self.g[L.bn + ".bias"].copy_(dbe[:L.cout])
            dz = d_out
        elif L.relu:
            dz = self._buf(L.key + ".dz", (mp, L.cp), torch.float16, zero=True)
            self._chk(lib.pcd_relu_mask_f16(d_out.data_ptr(), L.a.data_ptr(), m * L.cp, dz.data_ptr(), st), "relu_mask")
        else:
            dz = d_out
        if not L.bn:            # (in front of a BatchNorm the bias gradient is the analytic zero: dz is mean-free per channel)
            db = self._buf("bwd.db", (512,), torch.float32)
            self._chk(lib.pcd_colsum_f16(dz.data_ptr(), m, 1, L.cp, db.data_ptr(), st), "colsum")
            self.g[L.key + ".bias"].copy_(db[:L.cout])
        # dWm = dz^T col  (rows Cp, reduction Mp, columns Kp)
        dzT = self._scratch("bwd.dzT", L.cp * mp)
        colT = self._scratch("bwd.colT", L.kp * mp)
        self._chk(lib.pcd_transpose_f16(dz.data_ptr(), mp, L.cp, dzT.data_ptr(), st), "transpose")
        self._chk(lib.pcd_transpose_f16(L.col.data_ptr(), mp, L.kp, colT.data_ptr(), st), "transpose")
        dwm = self._buf(L.key + ".dwm", (L.cp, L.kp), torch.float32)
        g = _lib.GemmDesc()
        g.a1, g.lda1, g.k1 = dzT.data_ptr(), mp, mp
        g.w, g.ldw = colT.data_ptr(), mp
        g.m, g.c = L.cp, L.kp
        tiles = -(-L.cp // 128) * -(-L.kp // 128)
        splits = 1
        while splits * tiles < 512 and (mp // 64) % (splits * 2) == 0 and mp // (splits * 2) >= 256:
            splits *= 2
        if splits == 1:
            self._chk(lib.pcd_gemm_f16_out32(C.byref(g), dwm.data_ptr(), L.kp, st), "gemm_dW")
        else:
            slabs = self._scratch("bwd.slabs", splits * L.cp * L.kp, torch.float32)
            self._chk(lib.pcd_gemm_f16_splitk(C.byref(g), splits, slabs.data_ptr(), st), "gemm_dW_splitk")
            self._chk(lib.pcd_sum_slabs_f32(slabs.data_ptr(), splits, L.cp, L.kp, dwm.data_ptr(), L.kp, st), "sum_slabs")
        gw = dwm[:L.cout, :L.kk].reshape(L.cout, L.k, L.k, L.k, L.cin)
        self.g[L.key + ".weight"].copy_(gw.permute(4, 0, 1, 2, 3) if L.transposed else gw.permute(0, 4, 1, 2, 3))
        if not need_dx:
            return None
        # dcol = dz Wm, then the adjoint of the gather
        dcol = self._scratch("bwd.dcol", mp * L.kp)
        g2 = _lib.GemmDesc()
        g2.a1, g2.lda1, g2.k1 = dz.data_ptr(), L.cp, L.cp
        g2.w, g2.ldw = self.wmt[L.key].data_ptr(), L.cp
        g2.m, g2.c = mp, L.kp
        self._chk(lib.pcd_gemm_f16(C.byref(g2), dcol.data_ptr(), L.kp, st), "gemm_dcol")
        m_in = b * L.din ** 3
        dx = self._buf(L.key + ".dx", (_up(m_in, 64), L.cin), torch.float16, zero=True)
        self._chk(lib.pcd_col2im_f16(dcol.data_ptr(), b, L.cin, L.din, L.din, L.din, L.dout, L.dout, L.dout, L.k, L.s, L.p, L.transposed,
                                     L.kp, dx.data_ptr(), st), "col2im")
        return dx

    # (activations between layers are [Mp][Cp]; the next layer gathers only its Cin real channels: row stride must be Cin)
    def _narrow(self, a: torch.Tensor, L: _VConv, key: str) -> torch.Tensor:
        if L.cp == L.cout:
            return a
        out = self._buf(key, (a.shape[0], L.cout), torch.float16)
        out.copy_(a[:, :L.cout])
        return out

    def _widen(self, d: torch.Tensor, L: _VConv, key: str) -> torch.Tensor:
        """gradient wrt the narrowed output [.., Cout] -> [Mp][Cp] with zero padding channels."""
        if L.cp == L.cout and d.shape[0] == L.mp:
            return d
        out = self._buf(key, (L.mp, L.cp), torch.float16, zero=True)
        out[:d.shape[0], :L.cout] = d[:, :L.cout]
        return out

    def _run_fwd(self, ops, a: torch.Tensor, b: int, update_stats: bool) -> torch.Tensor:
        lib, st = self.lib, self._st()
        for op in ops:
            if op[0] == "conv":
                L = op[1]
                a = self._narrow(self._conv_fwd(L, a, b, update_stats), L, L.key + ".an")
            else:
                _, c1, c2, dn = op
                x_in = a
                h = self._narrow(self._conv_fwd(c1, x_in, b, update_stats), c1, c1.key + ".an")
                y = self._conv_fwd(c2, h, b, update_stats)
                r = self._conv_fwd(dn, x_in, b, update_stats) if dn is not None else self._widen(x_in, c2, c2.key + ".resw")
                out = self._buf(c2.key + ".out", (c2.mp, c2.cp), torch.float16, zero=True)
                self._chk(lib.pcd_add_relu_f16(y.data_ptr(), r.data_ptr(), c2.m * c2.cp, 1, out.data_ptr(), st), "add_relu")
                c2.out = out
                a = self._narrow(out, c2, c2.key + ".outn")
        return a

    def _run_bwd(self, ops, d: torch.Tensor, b: int, first_needs_dx: bool) -> Optional[torch.Tensor]:
        """d: gradient wrt the program's output, [rows][Cout of the last op]."""
        lib, st = self.lib, self._st()
        for n, op in enumerate(reversed(ops)):
            is_first = n == len(ops) - 1
            need_dx = first_needs_dx or not is_first
            if op[0] == "conv":
                L = op[1]
                d = self._conv_bwd(L, self._widen(d, L, L.key + ".dw"), b, need_dx)
            else:
                _, c1, c2, dn = op
                dw = self._widen(d, c2, c2.key + ".dw")
                dm = self._buf(c2.key + ".dmask", (c2.mp, c2.cp), torch.float16, zero=True)
                self._chk(lib.pcd_relu_mask_f16(dw.data_ptr(), c2.out.data_ptr(), c2.m * c2.cp, dm.data_ptr(), st), "relu_mask")
                d_res = dm.clone() if dn is None else dm            # bn backward rewrites its input in place
                dh = self._conv_bwd(c2, dm if dn is None else dm.clone(), b, True)
                dx_a = self._conv_bwd(c1, self._widen(dh, c1, c1.key + ".dw"), b, True)
                dx_b = self._conv_bwd(dn, d_res, b, True) if dn is not None else d_res
                rows = b * c1.din ** 3
                out = self._buf(c1.key + ".dxsum", (_up(rows, 64), c1.cin), torch.float16, zero=True)
                src_b = dx_b if dn is not None else self._narrow(dx_b, c2, c2.key + ".dresn")
                self._chk(lib.pcd_add_relu_f16(dx_a.data_ptr(), src_b.data_ptr(), rows * c1.cin, 0, out.data_ptr(), st), "add")
                d = out
        return d

    # ------------------------------------------------------------------ forward / backward / step
    def forward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, update_stats: bool = True):
        """(reconstruction (B,1,32,32,32) fp32, mu, logvar) with the module in train() mode."""
        lib, st = self.lib, self._st()
        b = x.shape[0]
        self.b = b
        self.x = x.to(self.dev, torch.float32).contiguous()
        a0 = self.x.reshape(b * 32768, 1).to(torch.float16)                     # C = 1: NCDHW and NDHWC coincide
        h = self._run_fwd(self.enc, a0, b, update_stats)                         # [64 (B rows valid)][512]
        self.h32 = self._buf("h32", (b, 512), torch.float32)
        self._chk(lib.pcd_f16_to_f32(h.data_ptr(), self.h32.data_ptr(), b * 512, st), "f16_to_f32")
        p = self.p
        self.mu = self._buf("mu", (b, 256), torch.float32)
        self.logvar = self._buf("logvar", (b, 256), torch.float32)
        self._mm(self.h32.data_ptr(), 512, 0, p["fc_mu.weight"].data_ptr(), 512, 1, b, 256, 512, p["fc_mu.bias"].data_ptr(), 0, self.mu.data_ptr(), 256)
        self._mm(self.h32.data_ptr(), 512, 0, p["fc_logvar.weight"].data_ptr(), 512, 1, b, 256, 512, p["fc_logvar.bias"].data_ptr(), 0,
                 self.logvar.data_ptr(), 256)
        if eps is None:
            eps = torch.empty(b, 256, device=self.dev)
            self._eps_offset = getattr(self, "_eps_offset", 0)       # running Philox position: every draw (training or validation) is fresh
            self._chk(lib.pcd_randn(eps.data_ptr(), eps.numel(), int(torch.initial_seed()) & (2 ** 64 - 1), (1 << 42) + self._eps_offset, st),
                      "randn")
            self._eps_offset += (eps.numel() + 3) // 4
        self.eps_draw = eps.to(self.dev, torch.float32).contiguous()
        self.zlat = self._buf("z", (b, 256), torch.float32)
        self._chk(lib.pcd_reparameterize(self.mu.data_ptr(), self.logvar.data_ptr(), self.eps_draw.data_ptr(), self.zlat.data_ptr(), b * 256, st),
                  "reparameterize")
        di = self._buf("dec_in", (b, 32768), torch.float32)
        self._mm(self.zlat.data_ptr(), 256, 0, p["decoder_input.weight"].data_ptr(), 256, 1, b, 32768, 256, p["decoder_input.bias"].data_ptr(), 0,
                 di.data_ptr(), 32768)
        a = di.view(b, 512, 64).transpose(1, 2).reshape(b * 64, 512).to(torch.float16)       # (B,512,4,4,4) -> channels-last rows
        logits = self._run_fwd(self.dec, a, b, update_stats)                     # last conv: [Mp][64], column 0 = the logit
        self.logits = self.dec[-1][1].a
        return None

    def backward(self, kl_weight: float):
        """BCE + kl_weight * KL and all parameter gradients (scaled by loss_scale).  Returns (loss, recon_loss, kl) tensors."""
        lib, st = self.lib, self._st()
        b, p, g = self.b, self.p, self.g
        last = self.dec[-1][1]
        n = b * 32768
        loss_sum = self._buf("loss", (1,), torch.float32)
        self.recon = self._buf("recon", (b, 1, 32, 32, 32), torch.float32)
        dl = self._buf(last.key + ".dlogit", (last.mp, last.cp), torch.float16, zero=True)
        self._chk(lib.pcd_sigmoid_bce(self.logits.data_ptr(), last.cp, self.x.data_ptr(), n, self.loss_scale, loss_sum.data_ptr(),
                                      self.recon.data_ptr(), dl.data_ptr(), st), "sigmoid_bce")
        # decoder: the last conv has no activation of its own (Sigmoid is in the loss kernel)
        d = self._conv_bwd(last, dl, b, True)
        d = self._run_bwd(self.dec[:-1], d, b, True)                             # [B*64][512] wrt the reshaped decoder input
        ddi = d[:b * 64].float().view(b, 64, 512).transpose(1, 2).reshape(b, 32768).contiguous()
        ones = self._buf("ones", (1, b), torch.float32)
        ones.fill_(1.0)
        self._mm(ddi.data_ptr(), 32768, 1, self.zlat.data_ptr(), 256, 0, 32768, 256, b, None, 0, g["decoder_input.weight"].data_ptr(), 256)
        self._mm(ones.data_ptr(), b, 0, ddi.data_ptr(), 32768, 0, 1, 32768, b, None, 0, g["decoder_input.bias"].data_ptr(), 32768)
        dz = self._buf("dz", (b, 256), torch.float32)
        self._mm(ddi.data_ptr(), 32768, 0, p["decoder_input.weight"].data_ptr(), 256, 0, b, 256, 32768, None, 0, dz.data_ptr(), 256)
        dmu, dlv = self._buf("dmu", (b, 256), torch.float32), self._buf("dlv", (b, 256), torch.float32)
        kl_sum = self._buf("kl", (1,), torch.float32)
        self._chk(lib.pcd_vae_latent_backward(self.mu.data_ptr(), self.logvar.data_ptr(), self.eps_draw.data_ptr(), dz.data_ptr(), b * 256,
                                              float(kl_weight) * self.loss_scale, dmu.data_ptr(), dlv.data_ptr(), kl_sum.data_ptr(), st),
                  "latent_backward")
        dh = self._buf("dh", (b, 512), torch.float32)
        for name, dd, acc in (("fc_mu", dmu, 0), ("fc_logvar", dlv, 1)):
            self._mm(dd.data_ptr(), 256, 1, self.h32.data_ptr(), 512, 0, 256, 512, b, None, 0, g[name + ".weight"].data_ptr(), 512)
            self._mm(ones.data_ptr(), b, 0, dd.data_ptr(), 256, 0, 1, 256, b, None, 0, g[name + ".bias"].data_ptr(), 256)
            self._mm(dd.data_ptr(), 256, 0, p[name + ".weight"].data_ptr(), 512, 0, b, 512, 256, None, acc, dh.data_ptr(), 512)
        d = dh.to(torch.float16)
        self._run_bwd(self.enc, d, b, False)
        recon_loss = loss_sum[0] / float(n)
        kl = -0.5 * kl_sum[0] / float(b * 256)
        return recon_loss + float(kl_weight) * kl, recon_loss, kl

    def grads(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() / self.loss_scale for k, v in self.g.items()}

    def optimizer_step(self):
        """torch.optim.Adam(lr) (networks.py:2290) = the AdamW kernel with zero weight decay."""
        self.step_count += 1
        b1, b2 = self.betas
        world = _allreduce_gradients(self.G)
        self._chk(self.lib.pcd_adamw_step(self.P.data_ptr(), self.G.data_ptr(), self.M1.data_ptr(), self.M2.data_ptr(), self.P.numel(),
                                          self.lr, b1, b2, self.eps, 0.0, self.step_count, self.loss_scale * world, self._st()), "adam")
        self.refresh_weights()

    step = optimizer_step

    def zero_grad(self):
        pass

    def train_step(self, x, kl_weight: float, eps=None):
        self.forward(x, eps)
        out = self.backward(kl_weight)
        self.optimizer_step()
        return out
