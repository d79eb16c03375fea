"""Parity of the layer-level HIP entry points (called through the C ABI) against the
CPU oracle / exact fp32-fp64 references.  GPU only."""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    from shapegen_amd import ops as o, _lib
    _lib.require_gpu()
    return o


def _int_mat(rows, cols, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, (rows, cols), generator=g).float()


@pytest.mark.parametrize("m,k,c", [(128, 64, 64), (256, 128, 128), (200, 64, 72), (1000, 256, 136), (64, 4096, 1024)])
def test_gemm_exact_integers(ops, m, k, c):
    """Small-integer operands make fp16 x fp16 -> fp32 exact: any fragment-layout mistake
    (swapped row/col, wrong k order, bad swizzle) shows up as a wrong integer.  A and W are
    asymmetric on purpose."""
    a, w = _int_mat(m, k, 1), _int_mat(c, k, 2)
    bias = _int_mat(1, c, 3).reshape(-1)
    want = a.double() @ w.double().t() + bias.double()
    got = ops.gemm_f16_out32(a.half().cuda(), w.half().cuda(), bias.cuda()).cpu().double()
    assert torch.equal(got, want)
    got16 = ops.gemm_f16(a.half().cuda(), w.half().cuda(), bias.cuda(), relu=True).cpu().double()
    assert torch.equal(got16, want.clamp_min(0).half().double())


@pytest.mark.parametrize("m,k,c", [(32768, 64, 512), (65536, 64, 256), (131072, 64, 512), (32768, 128, 4096),
                                   (4096, 64, 1024), (16384 + 256, 256, 512)])
def test_gemm_persistent_tile_mappings_exact(ops, m, k, c):
    """Large-M shapes that take the persistent 256-block grid with the XCD patch mapping (1, 2, 4, 8, 16
    column tiles; one and several rounds; a ragged last round): every output element must still be
    produced exactly once and from the right operands."""
    g = torch.Generator(device="cuda").manual_seed(m + c)
    a = torch.randint(-3, 4, (m, k), generator=g, device="cuda").half()
    w = torch.randint(-2, 3, (c, k), generator=g, device="cuda").half()
    bias = torch.randint(-3, 4, (c,), generator=g, device="cuda").float()
    want = (a.float() @ w.float().t() + bias).clamp_min(0)        # exact: small integers
    got = ops.gemm_f16(a, w, bias, relu=True).float()
    assert torch.equal(got, want)
    cm = ops.gemm_f16_colmax(a, w, bias, 2048 if m % 2048 == 0 else m)
    want_cm = want.reshape(-1, 2048 if m % 2048 == 0 else m, c).max(1)[0]
    assert torch.equal(cm, want_cm)


@pytest.mark.parametrize("depth_cfg", [7, 6])
def test_gemm_cross_tile_prefetch_kernel_exact(ops, depth_cfg):
    """gemm_xp_kernel (whole 256 x 256 tiles; the next tile's first K tile(s) requested before the epilogue's stores, bias by LDS-DMA,
    counted vmcnt waits): several tiles per workgroup, two K sources, a per-shape bias, no bias at all, no ReLU, the residual and the
    column-max epilogues -- exact on small integers, and bitwise equal to the generic kernel."""
    from shapegen_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    try:
        for m, k1, k2, c, use_bias, use_sb, relu in [(65536, 256, 0, 512, True, False, True), (32768, 128, 128, 1024, True, True, True),
                                                     (65536, 192, 0, 256, False, False, False), (16384 + 512, 192, 192, 768, False, True, True)]:
            a1 = torch.randint(-3, 4, (m, k1), generator=g, device="cuda").half()
            a2 = torch.randint(-3, 4, (m, k2), generator=g, device="cuda").half() if k2 else None
            w = torch.randint(-2, 3, (c, k1 + k2), generator=g, device="cuda").half()
            bias = torch.randint(-3, 4, (c,), generator=g, device="cuda").float() if use_bias else None
            rps = 256 if m % 2048 else 2048
            sb = torch.randint(-3, 4, (m // rps, c), generator=g, device="cuda").float() if use_sb else None
            want = (torch.cat([a1, a2], 1) if k2 else a1).float() @ w.float().t()
            if use_bias: want = want + bias
            if use_sb: want = want + sb.repeat_interleave(rps, 0)
            if relu: want = want.clamp_min(0)
            outs = {}
            for cfg in (depth_cfg, 5):
                lib.pcd_gemm_set_config(cfg)
                d = ops._desc(a1, w, bias, a2, sb, rps if use_sb else 0, relu)      # (two sources: equal row strides, the kernel's condition)
                out = torch.empty(m, c, dtype=torch.float16, device="cuda")
                _lib.check(lib.pcd_gemm_f16(d, out.data_ptr(), c, _lib.stream_ptr()), "gemm_f16")
                outs[cfg] = out
            assert torch.equal(outs[depth_cfg].float(), want.half().float()), (m, k1, k2, c)
            assert torch.equal(outs[depth_cfg], outs[5])
        # residual epilogue (the attention blocks' out_proj / FFN.2): out = fp16(fp16(acc + bias) + resid), 2 tiles per workgroup
        a = torch.randint(-3, 4, (65536, 256), generator=g, device="cuda").half()
        w = torch.randint(-2, 3, (512, 256), generator=g, device="cuda").half()
        bias = torch.randint(-3, 4, (512,), generator=g, device="cuda").float()
        resid = torch.randint(-5, 6, (65536, 512), generator=g, device="cuda").half()
        want = ((a.float() @ w.float().t() + bias).half().float() + resid.float()).half()
        res = {}
        for cfg in (depth_cfg, 5):
            lib.pcd_gemm_set_config(cfg)
            res[cfg] = ops.gemm_f16_residual(a, w, bias, resid)
        assert torch.equal(res[depth_cfg], want) and torch.equal(res[depth_cfg], res[5])
        # column max, several tiles per workgroup (65536 x 512 / 256^2 = 512 tiles)
        a = torch.randint(-3, 4, (65536, 256), generator=g, device="cuda").half()
        w = torch.randint(-2, 3, (512, 256), generator=g, device="cuda").half()
        bias = torch.randint(-3, 4, (512,), generator=g, device="cuda").float()
        want = (a.float() @ w.float().t() + bias).clamp_min(0).reshape(-1, 2048, 512).max(1)[0]
        for cfg in (depth_cfg, 5):
            lib.pcd_gemm_set_config(cfg)
            assert torch.equal(ops.gemm_f16_colmax(a, w, bias, 2048), want), cfg
    finally:
        lib.pcd_gemm_set_config(7)


def test_gemm_colmax_with_fragment_order_weights_exact(ops):
    """gemm_xw_kernel (global_feat.3's kernel: activation panel through LDS, the weights a fragment-order copy read straight from global memory into
    the MFMA operand registers, counted vmcnt over LDS-DMA pieces, asm loads and the epilogue's atomics): exact on small integers and bitwise equal
    to the LDS-staged kernel -- two-K-tile problems, long K, one / two / four output tiles per workgroup, two K sources; every launch repeated
    (a staging race would show as a column that differs); and the U-Net forward is bit-identical with the path on and off."""
    from shapegen_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(11)
    for m, k1, k2, c in [(65536, 128, 0, 256), (65536, 256, 0, 512), (32768, 128, 128, 1024), (16384, 2048, 0, 4096), (131072, 192, 0, 512)]:
        a1 = torch.randint(-3, 4, (m, k1), generator=g, device="cuda").half()
        a2 = torch.randint(-3, 4, (m, k2), generator=g, device="cuda").half() if k2 else None
        w = torch.randint(-2, 3, (c, k1 + k2), generator=g, device="cuda").half()
        bias = torch.randint(-3, 4, (c,), generator=g, device="cuda").float()
        want = ((torch.cat([a1, a2], 1) if k2 else a1).float() @ w.float().t() + bias).clamp_min(0).reshape(-1, 2048, c).max(1)[0]
        wfrag = torch.empty_like(w)
        _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), k1 + k2, k1 + k2, c, wfrag.data_ptr(), _lib.stream_ptr()))
        d = ops._desc(a1, w, bias, a2, relu=True)
        for rep in range(3):
            out = torch.zeros(m // 2048, c, dtype=torch.float32, device="cuda")
            _lib.check(lib.pcd_gemm_f16_colmax_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()), "gemm_f16_colmax_wfrag")
            assert torch.equal(out, want), (m, k1, k2, c, rep)
        assert torch.equal(out, ops.gemm_f16_colmax(a1, w, bias, 2048) if not k2 else out)
    # shapes the kernel does not take are refused, not mis-launched: fewer than 256 tiles
    a = torch.zeros(8192, 128, dtype=torch.float16, device="cuda")
    w = torch.zeros(256, 128, dtype=torch.float16, device="cuda")
    d = ops._desc(a, w, torch.zeros(256, device="cuda"), relu=True)
    out = torch.zeros(4, 256, device="cuda")
    assert lib.pcd_gemm_f16_colmax_wfrag(d, w.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()) != 0
    # the U-Net forward with the path on (default) and off
    from shapegen_amd.diffusion import PointCloudDiffusion
    from helpers import point_sd
    model = PointCloudDiffusion(num_points=2048)
    model.load_state_dict(point_sd(), strict=True)
    model = model.to("cuda").eval()
    x, t = torch.randn(16, 2048, 3, generator=torch.Generator().manual_seed(3)).cuda(), torch.rand(16, generator=torch.Generator().manual_seed(4)).cuda()
    assert lib.pcd_gemm_wfrag_enabled() == 1
    on = model.model(x, t).clone()
    lib.pcd_gemm_set_config(8)
    try:
        off = model.model(x, t).clone()
    finally:
        lib.pcd_gemm_set_config(9)
    assert torch.equal(on, off)


def test_gemm_store_with_fragment_order_weights_exact(ops):
    """gemm_xs_kernel (round 5: the store GEMMs of the point U-Net with K >= 512: weights straight from global memory, 5 / 6 of a wave's sixteen 1-KB output
    chunks stored from the epilogue, 11 / 10 parked in LDS and stored one / two per K tile of the NEXT output tile, every vmcnt wait counted over LDS-DMA
    pieces, asm weight loads and asm stores): exact on small integers and bitwise pcd_gemm_f16's output -- both drip rates (nk >= 12 and 6 <= nk < 12), one
    / two / eight output tiles per workgroup (the last tile of a workgroup stores all sixteen chunks itself), two K sources, per-shape bias, no bias, no
    ReLU; every launch repeated (a miscounted wait or a drip slot overwritten early shows as rows that differ between launches or from the reference)."""
    from shapegen_amd import _lib
    lib = _lib.load()
    assert lib.pcd_gemm_store_wfrag_enabled() == 1
    g = torch.Generator(device="cuda").manual_seed(12)
    cases = [  # m, k1, k2, c, rows_per_shape of a per-shape bias (0: none), bias, relu
        (65536, 1024, 0, 256, 0, True, True),         # 256 tiles: one per workgroup (all chunks direct), R = 1
        (65536, 1024, 0, 512, 2048, True, True),      # 512 tiles: the first tile drips, per-shape bias
        (32768, 512, 512, 2048, 0, True, True),       # two K sources, 1024 tiles: four per workgroup
        (131072, 512, 0, 1024, 0, True, True),        # R = 2 (8 K tiles), eight tiles per workgroup
        (65536, 384, 0, 512, 0, False, False),        # R = 2 at its smallest K (6 K tiles), no bias, no ReLU: negative outputs survive
        (65536, 768, 0, 512, 512, True, False),       # nk = 12: R = 1 at its smallest K
    ]
    for m, k1, k2, c, rps, has_bias, relu in cases:
        a1 = torch.randint(-3, 4, (m, k1), generator=g, device="cuda").half()
        a2 = torch.randint(-3, 4, (m, k2), generator=g, device="cuda").half() if k2 else None
        w = torch.randint(-2, 3, (c, k1 + k2), generator=g, device="cuda").half()
        bias = torch.randint(-3, 4, (c,), generator=g, device="cuda").float() if has_bias else None
        sb = torch.randint(-3, 4, (m // rps, c), generator=g, device="cuda").float() if rps else None
        wfrag = torch.empty_like(w)
        _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), k1 + k2, k1 + k2, c, wfrag.data_ptr(), _lib.stream_ptr()))
        d = ops._desc(a1, w, bias, a2, sb, rps, relu=relu)
        ref = ops.gemm_f16(a1, w, bias, a2, sb, rps, relu)                      # gemm_xp_kernel / the generic kernel
        pick = torch.cat([torch.arange(0, 512), torch.arange(m // 2 - 256, m // 2 + 256), torch.arange(m - 512, m)]).cuda()
        want = (torch.cat([a1, a2], 1) if k2 else a1)[pick].float() @ w.float().t()
        if has_bias:
            want = want + bias
        if rps:
            want = want + sb[pick // rps]
        want = (want.clamp_min(0) if relu else want).clamp(-65504, 65504).half()
        assert torch.equal(ref[pick], want), (m, k1, k2, c, "reference kernel")
        for rep in range(3):
            out = torch.full((m, c), float("nan"), dtype=torch.float16, device="cuda")
            _lib.check(lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), c, _lib.stream_ptr()), "gemm_f16_wfrag")
            assert torch.equal(out, ref), (m, k1, k2, c, rep, int((out != ref).any(1).sum()))
        if not relu:
            assert float(out.min()) < 0
    # random fp16 operands (every bit of the fp32 sums matters): bitwise the LDS-staged kernel, at the U-Net's widest store layer
    a = torch.randn(131072, 1024, generator=g, device="cuda").clamp_min(0).half()
    w = (torch.randn(2048, 1024, generator=g, device="cuda") / 32).half()
    bias = torch.randn(2048, generator=g, device="cuda") * 0.1
    wfrag = torch.empty_like(w)
    _lib.check(lib.pcd_gemm_pack_wfrag(w.data_ptr(), 1024, 1024, 2048, wfrag.data_ptr(), _lib.stream_ptr()))
    d = ops._desc(a, w, bias, relu=True)
    ref = ops.gemm_f16(a, w, bias, relu=True)
    out = torch.empty_like(ref)
    for rep in range(5):
        out.fill_(float("nan"))
        _lib.check(lib.pcd_gemm_f16_wfrag(d, wfrag.data_ptr(), out.data_ptr(), 2048, _lib.stream_ptr()))
        assert torch.equal(out, ref), rep
    # a shape the kernel does not take runs pcd_gemm_f16's kernels from the same entry point (fewer than 256 tiles; K = 128)
    a = torch.randint(-3, 4, (8192, 128), generator=g, device="cuda").half()
    w = torch.randint(-2, 3, (256, 128), generator=g, device="cuda").half()
    d = ops._desc(a, w, None, relu=True)
    out = torch.empty(8192, 256, dtype=torch.float16, device="cuda")
    _lib.check(lib.pcd_gemm_f16_wfrag(d, w.data_ptr(), out.data_ptr(), 256, _lib.stream_ptr()))
    assert torch.equal(out, ops.gemm_f16(a, w, None, relu=True))
    # the U-Net forward with the kernel on (default) and off: bit-identical
    from shapegen_amd.diffusion import PointCloudDiffusion
    from helpers import point_sd
    model = PointCloudDiffusion(num_points=2048)
    model.load_state_dict(point_sd(), strict=True)
    model = model.to("cuda").eval()
    x, t = torch.randn(16, 2048, 3, generator=torch.Generator().manual_seed(3)).cuda(), torch.rand(16, generator=torch.Generator().manual_seed(4)).cuda()
    on = model.model(x, t).clone()
    lib.pcd_gemm_set_config(10)
    try:
        assert lib.pcd_gemm_store_wfrag_enabled() == 0
        off = model.model(x, t).clone()
    finally:
        lib.pcd_gemm_set_config(11)
    assert torch.equal(on, off)


def test_gemm_dual_source_shape_bias_residual(ops):
    m, k1, k2, c, rps = 384, 128, 64, 136, 96
    a1, a2, w = _int_mat(m, k1, 4), _int_mat(m, k2, 5), _int_mat(c, k1 + k2, 6)
    sb = _int_mat(m // rps, c, 7)
    want = torch.cat([a1, a2], 1).double() @ w.double().t() + sb.double().repeat_interleave(rps, 0)
    got = ops.gemm_f16_out32(a1.half().cuda(), w.half().cuda(), None, a2=a2.half().cuda(),
                             shape_bias=sb.cuda(), rows_per_shape=rps).cpu().double()
    assert torch.equal(got, want)
    resid = _int_mat(m, c, 8)
    bias = _int_mat(1, c, 9).reshape(-1)
    w1 = _int_mat(c, k1, 10, -1, 2)
    want = (a1.double() @ w1.double().t() + bias.double()).half().double() + resid.double()
    got = ops.gemm_f16_residual(a1.half().cuda(), w1.half().cuda(), bias.cuda(), resid.half().cuda()).cpu().double()
    assert torch.equal(got, want.half().double())


@pytest.mark.parametrize("m,k1,k2,c", [(512, 64, 0, 64), (300, 64, 0, 128), (1000, 128, 128, 128), (256, 128, 0, 256)])
def test_gemm_hilo_weights(ops, m, k1, k2, c):
    """pcd_gemm_f16_hilo: W = hi + lo as two fp16 halves ([C][2 K]), the sources walked twice.  (1) Exact on integers with weights that
    NEED the second half (2049 = 2048 + 1 is not an fp16 number).  (2) On random fp32 weights the result follows the float64 product
    with the UNROUNDED weights ~500 times closer than the plain fp16-weight GEMM does."""
    g = torch.Generator().manual_seed(m + c)
    k = k1 + k2
    a = torch.randint(0, 3, (m, k), generator=g).float()
    w = torch.randint(-1, 2, (c, k), generator=g).float() * 2049.0
    hi = w.half()
    lo = (w.double() - hi.double()).half()
    assert not torch.equal(hi.double(), w.double()) and torch.equal(hi.double() + lo.double(), w.double())
    wh = torch.cat([hi, lo], 1).cuda().contiguous()
    a1, a2 = a[:, :k1].half().cuda().contiguous(), (a[:, k1:].half().cuda().contiguous() if k2 else None)
    scale = 1.0 / 64                                              # keep |out| inside fp16: fold a power of two into the inputs
    got = ops.gemm_f16_hilo((a1 * scale).half(), wh, None, a2=None if a2 is None else (a2 * scale).half(), relu=True).cpu().double()
    want = (a.double() * scale @ w.double().t()).clamp_min(0)
    assert float(want.max()) < 65504 and torch.equal(got, want.half().double())
    # random weights: error against the unrounded product
    a = torch.randn(m, k, generator=g).half()
    w = torch.randn(c, k, generator=g).double() / k ** 0.5
    hi = w.half()
    lo = (w - hi.double()).half()
    wh = torch.cat([hi, lo], 1).cuda().contiguous()
    a1, a2 = a[:, :k1].cuda().contiguous(), (a[:, k1:].cuda().contiguous() if k2 else None)
    exact = a.double() @ w.t()
    plain = ops.gemm_f16_out32(a1, hi.cuda().contiguous(), None, a2=a2).cpu().double()
    hilo32 = a.double() @ (hi.double() + lo.double()).t()
    got = ops.gemm_f16_hilo(a1, wh, None, a2=a2).cpu().double()
    e_plain = float((plain - exact).norm() / exact.norm())
    e_split = float((hilo32 - exact).norm() / exact.norm())
    assert e_plain > 1e-4 and e_split < 1e-6                      # what the second half buys, before the fp16 store
    assert float((got - hilo32.half().double()).abs().max()) <= 2 * float(torch.finfo(torch.float16).eps) * float(exact.abs().max())


@pytest.mark.parametrize("m,rps", [(512, 128), (512, 64), (480, 96), (300, 100)])
def test_gemm_colmax(ops, m, rps):
    k, c = 128, 200 if m == 300 else 256
    a, w = _int_mat(m, k, 11), _int_mat(c, k, 12)
    bias = _int_mat(1, c, 13).reshape(-1)
    full = (a.double() @ w.double().t() + bias.double()).clamp_min(0)
    want = torch.stack([full[i:i + rps].max(0)[0] for i in range(0, m, rps)])
    got = ops.gemm_f16_colmax(a.half().cuda(), w.half().cuda(), bias.cuda(), rps).cpu().double()
    assert torch.equal(got, want)


def test_gemm_random_fp16_accuracy(ops):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1024, 512, generator=g).half()
    w = (torch.randn(384, 512, generator=g) / 22).half()
    want = a.double() @ w.double().t()
    got = ops.gemm_f16_out32(a.cuda(), w.cuda()).cpu()
    assert rel_l2(got, want) < 2e-6   # fp32 accumulation of exact fp16 products


@pytest.mark.parametrize("shape,stride", [((3, 100, 3), 0), ((2, 64, 3), 1), ((5, 256), 1), ((1, 7, 3), 0)])
def test_ddpm_update_with_in_place_philox_draw(shape, stride):
    """pcd_ddpm_update_philox = pcd_randn_step + pcd_ddpm_update in one launch (the draw is never stored): bitwise the two launches, for
    shared and per-shape rates, sizes that are not multiples of 4, with and without the next state, in place."""
    from shapegen_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).cuda()
    eps = torch.randn(*shape, generator=g).cuda()
    B = shape[0]
    R = B if stride else 1
    rates = (torch.rand(4, R, generator=g) * 0.8 + 0.1).cuda()
    n, s_, co, s2 = (rates[i].contiguous() for i in range(4))
    counter = torch.tensor([6, 5], dtype=torch.int32, device="cuda")           # counter[1] = the step whose block is drawn
    seed, base, pstride = 0x1234567890ABCDEF, 1000, (x.numel() + 3) // 4
    per_shape = x.numel() // B
    z = torch.empty_like(x)
    _lib.check(lib.pcd_randn_step(z.data_ptr(), z.numel(), seed, base, pstride, counter.data_ptr(), _lib.stream_ptr()))
    for with_next in (True, False):
        x0a, xna = torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.pcd_ddpm_update(x.data_ptr(), eps.data_ptr(), z.data_ptr(), n.data_ptr(), s_.data_ptr(), co.data_ptr(), s2.data_ptr(),
                                       stride, x.numel(), per_shape, x0a.data_ptr(), xna.data_ptr() if with_next else 0, _lib.stream_ptr()))
        x0b, xin = torch.empty_like(x), x.clone()
        _lib.check(lib.pcd_ddpm_update_philox(xin.data_ptr(), eps.data_ptr(), n.data_ptr(), s_.data_ptr(), co.data_ptr(), s2.data_ptr(),
                                              stride, x.numel(), per_shape, x0b.data_ptr(), xin.data_ptr() if with_next else 0, seed, base,
                                              pstride, counter.data_ptr(), _lib.stream_ptr()))
        assert torch.equal(x0a, x0b)
        assert torch.equal(xin, xna if with_next else x)                       # in place when asked, untouched otherwise


def test_elementwise_updates_bit_exact(golden):
    """K4 kernels vs the reference's torch expressions: bit exact (fp contraction off)."""
    from shapegen_amd import _lib
    from oracle import torch_oracle as O
    lib = _lib.load()
    g = golden("point_unet.npz")
    x, eps, z = (torch.from_numpy(g[k]) for k in ("fw_small_x", "fw_small_eps", "step_z"))
    n, s, n2, s2 = [torch.full((2,), float(v)) for v in g["step_rates"]]
    n[1], s[1] = 0.7, 0.6   # per-shape rates (stride 1)
    st = _lib.stream_ptr()
    dx, de, dz = x.cuda(), eps.cuda(), z.cuda()
    dn, ds, dn2, ds2 = n.cuda(), s.cuda(), n2.cuda(), s2.cuda()
    x0, xn = torch.empty_like(dx), torch.empty_like(dx)
    _lib.check(lib.pcd_ddim_update(dx.data_ptr(), de.data_ptr(), dn.data_ptr(), ds.data_ptr(), dn2.data_ptr(),
                                   ds2.data_ptr(), 1, dx.numel(), 64 * 3, x0.data_ptr(), xn.data_ptr(), st))
    w_x0 = O.remove_noise(x, eps, n, s)
    w_xn = s2.view(-1, 1, 1) * w_x0 + n2.view(-1, 1, 1) * eps
    assert torch.equal(x0.cpu(), w_x0) and torch.equal(xn.cpu(), w_xn)
    coef = torch.sqrt(n2 / n)
    _lib.check(lib.pcd_ddpm_update(dx.data_ptr(), de.data_ptr(), dz.data_ptr(), dn.data_ptr(), ds.data_ptr(),
                                   coef.cuda().data_ptr(), ds2.data_ptr(), 1, dx.numel(), 64 * 3, x0.data_ptr(),
                                   xn.data_ptr(), st))
    w_xn = s2.view(-1, 1, 1) * w_x0 + coef.view(-1, 1, 1) * n.view(-1, 1, 1) * z
    assert torch.equal(xn.cpu(), w_xn)
    xt = torch.empty_like(dx)
    _lib.check(lib.pcd_add_noise(dx.data_ptr(), dz.data_ptr(), dn.data_ptr(), ds.data_ptr(), 1, dx.numel(), 192,
                                 xt.data_ptr(), st))
    assert torch.equal(xt.cpu(), s.view(-1, 1, 1) * x + n.view(-1, 1, 1) * z)
    _lib.check(lib.pcd_remove_noise(dx.data_ptr(), de.data_ptr(), dn.data_ptr(), ds.data_ptr(), 1, dx.numel(), 192,
                                    x0.data_ptr(), st))
    assert torch.equal(x0.cpu(), w_x0)


def test_randn_statistics():
    from shapegen_amd import _lib
    lib = _lib.load()
    out = torch.empty(1 << 20, device="cuda")
    _lib.check(lib.pcd_randn(out.data_ptr(), out.numel(), 24, 0, _lib.stream_ptr()))
    o = out.cpu().double()
    assert abs(o.mean()) < 5e-3 and abs(o.std() - 1) < 5e-3 and torch.isfinite(o).all()
    assert abs((o ** 4).mean() - 3) < 0.05
    out2 = torch.empty(1 << 20, device="cuda")
    _lib.check(lib.pcd_randn(out2.data_ptr(), out2.numel(), 24, 0, _lib.stream_ptr()))
    assert torch.equal(out, out2)                      # counter based: reproducible
    _lib.check(lib.pcd_randn(out2.data_ptr(), out2.numel(), 24, 1 << 18, _lib.stream_ptr()))
    assert not torch.equal(out, out2)


def test_metrics_kernels(golden):
    from shapegen_amd import metrics as M, utils as U
    from oracle import torch_oracle as O
    g = golden("metrics.npz")
    a, b = torch.from_numpy(g["m_a"]), torch.from_numpy(g["m_b"])
    assert torch.equal(M.normalize_to_cube(a.cuda()).cpu(), torch.from_numpy(g["m_norm_a"]))   # bit exact
    # Chamfer: HIP uses direct differences; the float64 direct evaluation is the yardstick,
    # the reference's matmul-form cdist agrees to ~1e-4 relative
    for x, y in ((a, b), (torch.from_numpy(g["units_x"]), torch.from_numpy(g["units_y"]))):
        got = float(M.chamfer_distance(x.cuda(), y.cuda(), 1))
        assert abs(got - float(O.chamfer_distance_exact(x, y, 1))) < 2e-6
        assert abs(got - float(O.chamfer_distance(x, y, 1))) < 1e-4          # north_star: |dCD| <= 1e-4
    assert abs(float(M.chamfer_distance(torch.from_numpy(g["units_x"]).cuda(), torch.from_numpy(g["units_y"]).cuda()))
               - float(g["units_cd"])) < 0.05
    assert float(M.chamfer_distance(a.cuda(), a.cuda(), 1)) == 0.0
    per = M.chamfer_per_sample(a.cuda(), b.cuda(), 1).cpu()
    for i in range(3):
        assert abs(float(per[i]) - float(O.chamfer_distance_exact(a[i], b[i], 1))) < 2e-6
    vox = U.voxelize(a.cuda()).cpu()
    assert torch.equal(vox, O.voxelize(a))                                    # integer work: bit exact
    for i in range(3):
        cd, emd, rec = M.compute_metrics(a[i].cuda(), b[i].cuda())
        w = g["m_triples"][i]
        assert abs(float(cd) - w[0]) < 0.1 and abs(float(emd) - w[1]) < 1e-4 * max(1, w[1]) and float(rec) == w[2]
    # ragged / empty inputs
    assert float(M.chamfer_distance(a[:, :7].cuda(), b[:, :200].cuda(), 1)) == pytest.approx(
        float(O.chamfer_distance_exact(a[:, :7], b[:, :200], 1)), abs=2e-6)


def test_voxels_to_points(golden):
    from shapegen_amd import utils as U
    from oracle import torch_oracle as O
    g = golden("latent.npz")
    dec = torch.from_numpy(g["vae_dec"])
    for thr in (0.4, 0.5):
        got = U.voxel_tensor_to_point_clouds(dec.cuda(), thr)
        for i, pc in enumerate(got):
            assert torch.equal(pc.cpu(), torch.from_numpy(g[f"v2p_thr{thr}_{i}"]))   # order + coords bit exact
    empty = torch.zeros(2, 1, 32, 32, 32)
    empty[1, 0, 31, 31, 31] = 1
    got = U.voxel_tensor_to_point_clouds(empty.cuda(), 0.5)
    assert got[0].shape == (0, 3) and torch.equal(got[1].cpu(), torch.ones(1, 3))
    full = torch.ones(1, 1, 32, 32, 32)
    got = U.voxel_tensor_to_point_clouds(full.cuda(), 0.5)[0].cpu()
    assert torch.equal(got, O.voxel_tensor_to_point_clouds(full, 0.5)[0])
    odd = (torch.rand(2, 1, 5, 7, 9, generator=torch.Generator().manual_seed(3)))
    for p, q in zip(U.voxel_tensor_to_point_clouds(odd.cuda(), 0.5), O.voxel_tensor_to_point_clouds(odd, 0.5)):
        assert torch.equal(p.cpu(), q)


def test_sinkhorn_emd(golden):
    """K12 vs the reference's `earth_mover_distance_gpu` goldens (units.py inputs: 5.9952).
    Tolerance 2e-3 relative: the reference's cost matrix comes from matmul-form cdist."""
    from shapegen_amd import metrics as M
    g = golden("metrics.npz")
    x, y = torch.from_numpy(g["units_x"]).cuda(), torch.from_numpy(g["units_y"]).cuda()
    got = float(M.earth_mover_distance_gpu(x, y))
    assert abs(got - float(g["units_emd_sinkhorn"])) < 2e-3 * float(g["units_emd_sinkhorn"])
    a, b = torch.from_numpy(g["m_a"]).cuda(), torch.from_numpy(g["m_b"]).cuda()
    got = float(M.earth_mover_distance_gpu(a, b))
    assert abs(got - float(g["m_emd_sinkhorn_batch"])) < 2e-3 * float(g["m_emd_sinkhorn_batch"])
    cd, emd, rec = M.compute_metrics(a[0], b[0], use_approximate_gpu_emd=True)
    w = g["m_triple_sinkhorn0"]
    assert abs(float(emd) - w[1]) < 2e-3 * w[1] and float(rec) == w[2] and abs(float(cd) - w[0]) < 0.1


def test_gemm_splitk_matches_single_pass():
    """pcd_gemm_f16_splitk + pcd_sum_slabs_f32 (backward-weight products: few output tiles, long reduction)."""
    import ctypes as C
    from shapegen_amd import _lib
    lib, st = _lib.load(), _lib.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(3)
    for m, c, k, splits in ((64, 72, 4096, 16), (300, 256, 2048, 4), (1024, 128, 8192, 8)):
        a = torch.randint(-3, 4, (m, k), device="cuda", generator=g).half()
        w = torch.randint(-3, 4, (c, k), device="cuda", generator=g).half()
        d = _lib.GemmDesc()
        d.a1, d.lda1, d.k1, d.w, d.ldw, d.m, d.c = a.data_ptr(), k, k, w.data_ptr(), k, m, c
        slabs = torch.empty(splits, m, c, device="cuda")
        _lib.check(lib.pcd_gemm_f16_splitk(C.byref(d), splits, slabs.data_ptr(), st))
        out = torch.full((m, c + 5), -7.0, device="cuda")
        _lib.check(lib.pcd_sum_slabs_f32(slabs.data_ptr(), splits, m, c, out.data_ptr(), c + 5, st))
        want = a.float() @ w.float().t()                       # small integers: exact in fp32
        assert torch.equal(out[:, :c], want) and bool((out[:, c:] == -7.0).all())
        for s_ in range(splits):
            ks = k // splits
            assert torch.equal(slabs[s_], a[:, s_ * ks:(s_ + 1) * ks].float() @ w[:, s_ * ks:(s_ + 1) * ks].float().t())


def test_pair_metrics_batched_equals_per_pair_calls():
    """`pcd_pair_metrics`: ragged pairs in one enqueue (per-pair normalisation, Chamfer with the target range split over
    blocks, Sinkhorn with the convergence test kept on the device, voxel BCE from bit sets) against the per-pair calls
    of compute_metrics and against the CPU oracle."""
    from shapegen_amd import metrics as M
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(12)
    sizes = [(300, 257), (64, 500), (5, 40), (2048, 2048), (777, 3)]
    a = [torch.rand(n, 3, generator=g) * 2 - 1 for n, _ in sizes]
    b = [a[i][: sizes[i][1]].clone() * 0.9 + 0.05 * torch.randn(min(sizes[i]), 3, generator=g) if sizes[i][1] <= sizes[i][0]
         else torch.rand(sizes[i][1], 3, generator=g) * 2 - 1 for i in range(len(sizes))]
    a.append(torch.zeros(0, 3)); b.append(torch.rand(5, 3, generator=g))          # an empty cloud: NaN row
    ac, bc = [t.cuda() for t in a], [t.cuda() for t in b]
    rows = M.pair_metrics(ac, bc, use_approximate_gpu_emd=True).cpu()
    assert rows.shape == (6, 3) and torch.isnan(rows[5]).all()
    for i in range(5):
        cd, emd, bce = M.compute_metrics(ac[i], bc[i], True)
        assert abs(float(rows[i, 0]) - float(cd)) <= 2e-6 * max(1.0, float(cd)), i
        assert abs(float(rows[i, 1]) - float(emd)) <= 2e-4 * max(1.0, float(emd)) + 1e-6, i
        assert float(rows[i, 2]) == float(bce), i
        want = O.compute_metrics(a[i], b[i], True)
        assert abs(float(rows[i, 0]) - float(want[0])) < 0.15 + 1e-4 * float(want[0])      # matmul-form cdist in the oracle (x1e3)
        assert abs(float(rows[i, 1]) - float(want[1])) <= 2e-3 * max(1.0, float(want[1])) + 1e-5
        assert float(rows[i, 2]) == float(want[2])
    # dense (P, N, 3) tensors take the same path; one pair spreads over the chip through the target split
    x, y = torch.rand(3, 512, 3, generator=g).cuda(), torch.rand(3, 512, 3, generator=g).cuda()
    dense = M.pair_metrics(x, y, True).cpu()
    for i in range(3):
        cd, emd, bce = M.compute_metrics(x[i], y[i], True)
        assert abs(float(dense[i, 0]) - float(cd)) <= 2e-6 * float(cd) and float(dense[i, 2]) == float(bce)
        assert abs(float(dense[i, 1]) - float(emd)) <= 2e-4 * float(emd)
    # exact (Hungarian) EMD column comes from the host solve, like the reference
    ex = M.pair_metrics(x[:1], y[:1], False).cpu()
    assert abs(float(ex[0, 1]) - float(M.earth_mover_distance_cpu(x[0], y[0]))) < 1e-6


def test_pair_metrics_at_the_evaluation_size_next_to_the_oracle():
    """`pair_metrics` at the size an evaluation of BASELINE configs[1] runs it (64 pairs of 2048 x 2048 points in one enqueue:
    Chamfer, Sinkhorn EMD, voxel BCE) against the ORACLE's per-pair `compute_metrics` on four of the pairs (first, two in the
    middle, last).  Chamfer: the oracle's matmul-form cdist is ~1e-4 (x1e3: 0.1) off the exact value."""
    from shapegen_amd import metrics as M
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(2048)
    a = torch.tanh(torch.randn(64, 2048, 3, generator=g))
    b = (a + 0.05 * torch.randn(64, 2048, 3, generator=g)).clamp(-1, 1)
    rows = M.pair_metrics(a.cuda(), b.cuda(), use_approximate_gpu_emd=True).cpu()
    assert rows.shape == (64, 3) and torch.isfinite(rows).all()
    for i in (0, 21, 42, 63):
        want = [float(v) for v in O.compute_metrics(a[i], b[i], True)]
        assert abs(float(rows[i, 0]) - want[0]) < 0.15 + 1e-4 * want[0], i
        assert abs(float(rows[i, 1]) - want[1]) <= 2e-3 * max(1.0, want[1]) + 1e-5, i
        assert float(rows[i, 2]) == want[2], i


def test_pair_metrics_c_entry_point_with_an_empty_cloud():
    """the C entry point itself (not only the Python wrapper, which filters such pairs) answers a zero count with a
    NaN row and leaves the other pairs untouched."""
    from shapegen_amd import _lib, metrics as M
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(3, 96, 3, generator=g).cuda(), torch.rand(3, 80, 3, generator=g).cuda()
    na = torch.tensor([96, 0, 50], dtype=torch.int32, device="cuda")
    nb = torch.tensor([80, 80, 0], dtype=torch.int32, device="cuda")
    log_mu = torch.log(1.0 / na.clamp_min(1).float().cpu() + 1e-10).cuda()
    log_nu = torch.log(1.0 / nb.clamp_min(1).float().cpu() + 1e-10).cuda()
    rows = torch.empty(3, 3, device="cuda")
    need = int(lib.pcd_pair_metrics_workspace_bytes(3, 96, 80))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    _lib.check(lib.pcd_pair_metrics(a.data_ptr(), na.data_ptr(), 96, b.data_ptr(), nb.data_ptr(), 80, 3, 1, 1e-2, 1e-5, 100,
                                    log_mu.data_ptr(), log_nu.data_ptr(), rows.data_ptr(), ws.data_ptr(), need, _lib.stream_ptr()))
    rows = rows.cpu()
    assert torch.isnan(rows[1]).all() and torch.isnan(rows[2]).all() and torch.isfinite(rows[0]).all()
    cd, emd, bce = M.compute_metrics(a[0], b[0], True)
    assert abs(float(rows[0, 0]) * 1e3 - float(cd)) <= 2e-6 * float(cd) and float(rows[0, 2]) == float(bce)
