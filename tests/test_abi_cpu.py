"""CPU-side checks of the boundary: the shared library loads, exports every symbol the
header declares, the ctypes table covers them, and the host logic that needs no GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pcd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcd_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from shapegen_amd import _lib
    _lib.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pcd_hip.h but not exported"
        assert n in _lib._SIGS, f"{n} has no ctypes prototype"
    assert set(_lib._SIGS) == set(names)
    assert lib.pcd_abi_version() == _lib.ABI_VERSION == 2


def test_arg_validation_without_gpu():
    """Argument errors are reported before any device work."""
    from shapegen_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.pcd_gemm_f16(d, 0, 0, 0) == -1
    assert b"bad argument" in lib.pcd_last_error()
    assert lib.pcd_unet_workspace_bytes(64, 2048) == lib.pcd_unet_workspace_bytes(64, 2048) > 1 << 30
    assert lib.pcd_unet_workspace_bytes(0, 5) == 0
    # conv3d: shape logic is host side -- the LDS-halo path is offered only for k3 / s1 / p1, C_in 32 or 64 and
    # volumes that tile by (4, 4, 8); split-K scratch is asked for exactly when the launch would be small
    c = _lib.Conv3dDesc()
    assert lib.pcd_conv3d_k3s1_supported(c) == 0 and lib.pcd_conv3d_workspace_bytes(c, 1) == 0
    assert lib.pcd_conv3d_f16(c, 0) == -1 and lib.pcd_conv3d_k3s1_f16(c, 0) == -1
    c.inp = c.w = c.out = c.taps = c.zero_page = 64                       # never dereferenced on the host
    c.batch, c.in_d, c.in_h, c.in_w, c.cin, c.cout = 32, 32, 32, 32, 64, 64
    c.rows_d = c.rows_h = c.rows_w = c.out_d = c.out_h = c.out_w = 32
    c.stride, c.out_scale, c.ntaps, c.kpad = 1, 1, 27, 27 * 64
    assert lib.pcd_conv3d_k3s1_supported(c) == 1
    assert lib.pcd_conv3d_workspace_bytes(c, 1) == 0                         # 8192 output tiles: no split
    c.in_w = c.rows_w = c.out_w = 36
    assert lib.pcd_conv3d_k3s1_supported(c) == 0
    c.in_d = c.in_h = c.in_w = c.rows_d = c.rows_h = c.rows_w = c.out_d = c.out_h = c.out_w = 4
    c.cin, c.cout, c.kpad = 512, 512, 27 * 512                               # encoder.11: 16 x 4 tiles, 216 K tiles
    ws = lib.pcd_conv3d_workspace_bytes(c, 1)
    assert ws > 0 and ws % (32 * 64 * 512 * 4) == 0                          # whole fp32 slabs [M][C_out]
    assert lib.pcd_conv3d_workspace_bytes(c, 9) == 0                         # more than 8 variants: refused


def test_packing_is_exact_algebra():
    """BN folding + refine folding + hoisting reproduce the oracle forward in float64."""
    from shapegen_amd import packing, specs
    from oracle import torch_oracle as O
    from helpers import point_sd, rel_l2
    sd = point_sd("")
    lin, ex = packing.pack_point_unet(sd, "", 256, 256)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(2, 32, 3, generator=g), torch.tensor([0.3, 0.8])
    want = O.unet_pointnet_large(sd, "", x, t)
    temb = O.time_mlp(sd, "", O.timestep_embedding(t, 256)).double().numpy()
    relu = lambda v: np.maximum(v, 0)
    tb = temb @ ex["e1w_t"].T + ex["e1b"]
    h = relu(x.double().numpy() @ ex["e1w_xyz"].T + tb[:, None, :])
    f = lambda i, v: relu(v @ lin[i][0].T + lin[i][1])
    h = f(1, f(0, h)); x1 = h
    h = f(4, f(3, f(2, h))); x2 = h
    h = f(7, f(6, f(5, h))); x3 = h
    h = f(10, f(9, f(8, h))); x4 = h
    pooled = f(12, f(11, h)).max(axis=1)
    gb = pooled @ ex["wg"].T + lin[13][1]
    h = relu(x4 @ lin[13][0].T + gb[:, None, :])
    h = f(15, f(14, h))
    h = f(18, f(17, f(16, np.concatenate([h, x3], -1))))
    h = f(21, f(20, f(19, np.concatenate([h, x2], -1))))
    h = f(24, f(23, f(22, np.concatenate([h, x1], -1))))
    h = f(25, h)
    eps = h @ ex["head_w"].T + ex["head_b"]
    assert rel_l2(eps, want) < 1e-5
    with pytest.raises(RuntimeError):
        packing.pack_point_unet(sd, "", 256, 512)


def test_hi_lo_weight_split_and_the_forward_it_implies():
    """`packing.split_hilo`: [C][2 K] fp16 = hi | lo with hi + lo = w to ~2^-22; and a float64 emulation of the product path's WEIGHT
    precision on (2, 32): folded weights rounded to fp16 everywhere, against the same with the narrow layers (the library's
    PCD_UNET_HILO_ALLOWED mask) carried as hi + lo -- the second is closer to the oracle on eps (activations stay float64 here, so
    this isolates what the hi / lo weights change; the measured 1000-step effect is in profiles/r04_f)."""
    from shapegen_amd import _lib, packing
    from oracle import torch_oracle as O
    from helpers import point_sd, rel_l2
    g = np.random.default_rng(0)
    w = g.standard_normal((64, 128)) / 11.3
    hl = packing.split_hilo(w)
    assert hl.dtype == np.float16 and hl.shape == (64, 256)
    back = hl[:, :128].astype(np.float64) + hl[:, 128:].astype(np.float64)
    assert np.abs(back - w).max() <= 2.0 ** -21 * np.abs(w).max()
    assert np.abs(hl[:, :128].astype(np.float64) - w).max() > 2.0 ** -14 * np.abs(w).max()      # what plain fp16 rounding loses
    assert _lib.PCD_UNET_HILO_ALLOWED == sum(1 << i for i in (0, 1, 4, 22, 23, 24, 25))
    sd = point_sd("")
    lin, ex = packing.pack_point_unet(sd, "", 256, 256)
    gen = torch.Generator().manual_seed(1)
    x, t = torch.randn(2, 32, 3, generator=gen), torch.tensor([0.4, 0.7])
    want = O.unet_pointnet_large(sd, "", x, t)
    temb = O.time_mlp(sd, "", O.timestep_embedding(t, 256)).double().numpy()
    relu = lambda v: np.maximum(v, 0)
    r16 = lambda a: a.astype(np.float16).astype(np.float64)

    def forward(mask):
        def wq(i):
            if (mask >> i) & 1:
                h = packing.split_hilo(lin[i][0]).astype(np.float64)
                return h[:, :lin[i][0].shape[1]] + h[:, lin[i][0].shape[1]:]
            return r16(lin[i][0])
        f = lambda i, v: relu(v @ wq(i).T + lin[i][1])
        tb = temb @ ex["e1w_t"].T + ex["e1b"]
        h = relu(x.double().numpy() @ ex["e1w_xyz"].T + tb[:, None, :])
        h = f(1, f(0, h)); x1 = h
        h = f(4, f(3, f(2, h))); x2 = h
        h = f(7, f(6, f(5, h))); x3 = h
        h = f(10, f(9, f(8, h))); x4 = h
        pooled = f(12, f(11, h)).max(axis=1)
        gb = pooled @ r16(ex["wg"]).T + lin[13][1]
        h = relu(x4 @ wq(13).T + gb[:, None, :])
        h = f(15, f(14, h))
        h = f(18, f(17, f(16, np.concatenate([h, x3], -1))))
        h = f(21, f(20, f(19, np.concatenate([h, x2], -1))))
        h = f(24, f(23, f(22, np.concatenate([h, x1], -1))))
        return f(25, h) @ ex["head_w"].T + ex["head_b"]

    e_plain, e_hilo = rel_l2(forward(0), want), rel_l2(forward(_lib.PCD_UNET_HILO_ALLOWED), want)
    assert e_hilo < 0.6 * e_plain and e_plain < 3e-3, (e_plain, e_hilo)


def test_module_state_dict_and_cpu_failure():
    from shapegen_amd import specs
    from shapegen_amd.diffusion import PointCloudDiffusion
    m = PointCloudDiffusion(num_points=32)
    want = [(k, s) for k, s, _ in specs.unet_pointnet_large_spec(prefix="model.")]
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == want
    assert m.hparams.num_points == 32 and m.hparams["noise_schedule"] == "cosine"
    n, s = m.diffusion_schedule(torch.tensor([0.0, 0.5, 1.0]))
    np.testing.assert_allclose(s.numpy(), [0.95, 0.5944798, 0.02000003], rtol=1e-6)
    with pytest.raises(RuntimeError):
        m.sample(1, 32, num_steps=1)     # no CPU path: fail loudly


def test_linear_schedule_bug_for_bug():
    from shapegen_amd.diffusion import PointCloudDiffusion
    m = PointCloudDiffusion(num_points=8, noise_schedule="linear")
    n, s = m.diffusion_schedule(torch.tensor([0.5, 0.5, 0.5]))
    np.testing.assert_allclose(s.numpy(), [0.98995, 0.98000, 0.97015], atol=1e-5)   # SURVEY a2 probe


@pytest.mark.parametrize("T", [1, 2, 7, 100, 1000])
def test_vectorized_step_tables_equal_the_per_step_loop(T):
    """The cosine-schedule step tables are built with one set of elementwise ops over all T steps; the literal
    per-step transcription of the reference loops (diffusion.py:241-255, 277-286, 323-335) must give the same bits."""
    from shapegen_amd.diffusion import PointCloudDiffusion
    m = PointCloudDiffusion(num_points=8)
    builders = [lambda: m.ddim_table(T, 4), lambda: m.ddpm_table(T, 4), lambda: m.from_state_table(torch.tensor(0.37), T),
                lambda: m.from_state_table(1.0, T)]
    for fn in builders:
        m.vectorized_tables = True
        a = fn()
        m.vectorized_tables = False
        b = fn()
        assert a.steps == b.steps == T and a.width == b.width == 1 and a.stride == b.stride
        for f in ("t", "n", "s", "a", "b"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f


def test_persistent_latent_plan_is_a_partition():
    """csrc/latent_persist.hip's static work assignment, checked on the host without a device: every (layer, 32-column tile,
    64-k chunk) belongs to exactly one gemm unit, every (GroupNorm group, row) of a layer with partial slabs to exactly one
    finish unit, unit lists are in phase order and every workgroup's LDS plan fits 160 KB."""
    from shapegen_amd import _lib
    set_bytes = _lib.load().pcd_latent_persist_plan_check()
    assert set_bytes > 0 and set_bytes % 4096 == 0 and set_bytes < 8 << 20
