"""Multi-rank path on the GPU: two processes on one MI355X (gloo rendezvous, host-staged gathers) run
`dist.sample_sharded` / `dist.evaluate_sharded`, and `bench.py --gpus 2` launches its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def test_two_rank_sampling_and_evaluation_match_one_process(tmp_path):
    out = str(tmp_path / "dist.npz")
    port = _port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, DIST_OUT=out, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    r = np.load(out)
    assert int(r["world"]) == 2
    assert bool(r["sharded_equal"])            # injected x_T: bit-identical to the single-process sampler
    assert bool(r["drawn_equal"])              # on-device Philox noise: sample i is world-size invariant
    assert bool(r["drawn2_equal"])             # ... including the per-step DDPM draws
    assert bool(r["halves_differ"])            # and the two ranks did not draw the same stream
    assert r["rows"].shape == (6, 3)
    np.testing.assert_allclose(r["rows"], r["want_rows"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["mean"], r["want_rows"].mean(0), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("config", ["cfg2", "cfg3", "cfg5"])
def test_bench_launches_its_own_ranks(config):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts the ranks (PCD_BENCH_SHARE_GPU=1: both on
    this box's one GPU, collectives over gloo) and relays rank 0's JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PCD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    steps = ["--steps", "6", "--warmup", "2"] if config == "cfg2" else ["--steps", "50", "--warmup", "50"]   # cfg5 ignores both (T = 1000)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config,
                        "--no-cpu-baseline", "--no-attention"] + steps, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["rccl_ranks"] == 2 and rec["config"]["collective_backend"] == "gloo"
    roof = rec["roofline"]                                  # every config's line carries a live-timed dominant kernel
    assert roof["achieved"] > 0 and 0 < roof["frac"] < 1.5 and roof["avg_launch_ms"] > 0 and roof["bound"] in ("mfma", "hbm")
    if config == "cfg5":
        assert rec["steps"] == 1000 and rec["config"]["global_batch"] == 64 and len(rec["config"]["mean_metrics"]) == 3
        assert rec["roofline_vae_decode"]["bound"] == "mfma"


def _one_rank_env():
    env = {k: v for k, v in os.environ.items() if k not in ("PCD_BENCH_SHARE_GPU",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=_port(), PYTHONPATH=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0", PCD_DIST_FORCE_COLLECTIVE="1", PCD_COLLECTIVE_TIMEOUT_S="120")
    return env


def test_rccl_branch_on_one_gpu():
    """The production collective branch (backend "nccl" = RCCL, device tensors, `device_id=`) executed on the one GPU of
    this box: a one-rank world with PCD_DIST_FORCE_COLLECTIVE=1 takes no `world == 1` shortcut, so `all_gather_rows`
    (with and without counts), `all_gather_clouds` on ragged device clouds, `evaluate_sharded`, `sample_sharded` and
    `bench.Ranks.max_over_ranks / collective_ranks` all run through RCCL in a child process.  Not a scaling measurement."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_gpu_worker.py")], env=_one_rank_env(),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["backend"] == "nccl" and res["ranks_backend"] == "nccl" and res["world"] == 1 and res["host_staged"] is False
    for k in ("rows_equal", "clouds_equal", "eval_equal", "mean_ok", "sampler_equal"):
        assert res[k] is True, k
    assert res["max_over_ranks"] == 1.25 and res["collective_ranks"] == 1


def test_bench_cfg5_line_over_rccl_on_one_gpu():
    """`bench.py --config cfg5` as ONE rank over RCCL (forced): the all-gather of the ragged clouds and of the metric rows
    run on device tensors; the line says so (`collective_backend` "nccl", `rccl_ranks` 1)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "cfg5", "--no-cpu-baseline"],
                       env=_one_rank_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2500:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["config"]["rccl_ranks"] == 1 and rec["config"]["collective_backend"] == "nccl"
    assert rec["steps"] == 1000 and rec["config"]["global_batch"] == 32 and len(rec["config"]["mean_metrics"]) == 3


def test_bench_cfg4_line():
    """`python bench.py --config cfg4` (BASELINE configs[3] on one GPU): encode 32 grids, 1000 latent steps, decode,
    voxel -> points; the line carries the latent step against the HBM roofline and the VAE legs against the MFMA peak."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg4", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 1000 and rec["value"] > 1000
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["bytes_per_launch"] == 38174720.0
    assert abs(rec["roofline"]["achieved"] - 38174720.0 / (rec["config"]["latent_us_per_step"] * 1e-6) / 1e9) < 1e-6 * rec["roofline"]["achieved"]
    assert rec["roofline_vae_decode"]["frac"] > 0.05 and rec["roofline_vae_encode"]["frac"] > 0.05
    assert rec["config"]["cfg4_ms_end_to_end"] > rec["config"]["encode_ms"] + rec["config"]["decode_ms"]
