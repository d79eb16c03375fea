"""N > 1 path on CPU: world_size-2 (and one world_size-8) gloo processes exercise the sharding and the all-gather layer
(`shapegen_amd.dist`) that runs over RCCL on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the HIP sampler: the distributed layer only needs `.device` and `.sample`."""
    device = torch.device("cpu")

    def sample(self, n, num_points, num_steps=1, x_T=None):
        return x_T * 2.0 + num_steps


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shapegen_amd  # noqa: F401
    from shapegen_amd import dist as D
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # even and uneven shards
    assert D.shard_range(8, rank, world) == ((0, 4) if rank == 0 else (4, 8))
    assert D.shard_range(7, rank, world) == ((0, 4) if rank == 0 else (4, 7))
    g = torch.Generator().manual_seed(24)
    x_T = torch.randn(7, 16, 3, generator=g)
    out = D.sample_sharded(_FakeModel(), 7, 16, 5, x_T_global=x_T)
    assert torch.equal(out, x_T * 2.0 + 5)                       # rank-order concat == single-process result
    rows = torch.arange(3 * (rank + 1), dtype=torch.float32).reshape(rank + 1, 3) + 100 * rank
    allr = D.all_gather_rows(rows)
    assert allr.shape == (3, 3) and torch.equal(allr[0], torch.tensor([0., 1., 2.])) and allr[1, 0] == 100
    clouds = [torch.full((rank + 2 + i, 3), float(rank * 10 + i)) for i in range(2)]
    allc = D.all_gather_clouds(clouds)
    assert [c.shape[0] for c in allc] == [2, 3, 3, 4] and float(allc[3][0, 0]) == 11.0
    assert D.all_gather_clouds([torch.zeros(0, 3), torch.ones(1, 3)])[2 * rank].shape[0] == 0
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _worker_n(rank, world, port, tmp):
    """The sharding / gathering layer at the world size the driver's scaling run uses (8): uneven shards, ranks with NOTHING to do (fewer samples
    than ranks), ragged row and cloud gathers with empty contributions, the per-rank Philox sub-blocks."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shapegen_amd  # noqa: F401
    from shapegen_amd import dist as D
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    for total in (512, 256, 13, 5, 0):                       # BASELINE configs[2] / [4]'s batches; uneven; fewer samples than ranks; none
        spans = [D.shard_range(total, k, world) for k in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    g = torch.Generator().manual_seed(24)
    for total in (13, 5):
        x_T = torch.randn(total, 16, 3, generator=g)
        out = D.sample_sharded(_FakeModel(), total, 16, 5, x_T_global=x_T)
        assert torch.equal(out, x_T * 2.0 + 5)               # rank-order concat == the single-process result, empty shards included
    n_rows = rank % 3                                        # 0, 1, 2, 0, ... rows per rank
    rows = torch.full((n_rows, 4), float(rank))
    allr = D.all_gather_rows(rows)
    want = torch.cat([torch.full((k % 3, 4), float(k)) for k in range(world)])
    assert torch.equal(allr, want)
    clouds = [torch.full((rank + i, 3), float(10 * rank + i)) for i in range(rank % 2 + 1)]          # 1 or 2 clouds, the first of rank 0 empty
    allc = D.all_gather_clouds(clouds)
    flat = [(k + i, float(10 * k + i)) for k in range(world) for i in range(k % 2 + 1)]
    assert [c.shape[0] for c in allc] == [n for n, _ in flat]
    assert all(c.numel() == 0 or float(c[0, 0]) == v for c, (_, v) in zip(allc, flat))
    lo, hi = D.shard_range(13, rank, world)
    class _M:
        _shard = None
    from shapegen_amd.diffusion import _DiffusionBase
    m = _M()
    with D.shard_context(m, lo, 13):
        off, span = _DiffusionBase._philox_span(m, max(hi - lo, 1) * 64 * 3, max(hi - lo, 1))
    assert (off, span) == (lo * 48, 13 * 48)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"n_ok{rank}"), "w").write("ok")


def test_eight_rank_gloo(tmp_path):
    mp.spawn(_worker_n, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"n_ok{k}").exists() for k in range(8))


def _forced_one_rank_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shapegen_amd  # noqa: F401
    from shapegen_amd import dist as D
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PCD_DIST_FORCE_COLLECTIVE="1")
    assert D.init_from_env("gloo") == (0, 1, 0) and dist.is_initialized()
    t = torch.arange(6.).reshape(2, 3)
    out = D.all_gather_rows(t)
    assert out is not t and torch.equal(out, t)                  # went through the backend, not the world == 1 shortcut
    assert torch.equal(D.all_gather_rows(t, counts=[2]), t)
    clouds = [torch.ones(3, 3), torch.zeros(0, 3), torch.full((5, 3), 2.0)]
    got = D.all_gather_clouds(clouds)
    assert len(got) == 3 and all(torch.equal(a, b) for a, b in zip(got, clouds))
    dist.destroy_process_group()
    open(os.path.join(tmp, "forced_ok"), "w").write("ok")


def test_forced_collective_in_a_one_rank_world(tmp_path):
    """PCD_DIST_FORCE_COLLECTIVE=1: the switch that lets a one-GPU box execute the RCCL branch (tests/test_gpu_dist.py)
    covered on the CPU backend: a world of one rank initialises its group and every gather goes through the backend."""
    mp.spawn(_forced_one_rank_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "forced_ok").exists()


def test_single_process_passthrough():
    from shapegen_amd import dist as D
    assert D.world() == (0, 1)
    t = torch.arange(6.).reshape(2, 3)
    assert D.all_gather_rows(t) is t
    assert D.shard_range(5, 0, 1) == (0, 5)


def _eval_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shapegen_amd  # noqa: F401
    from shapegen_amd import dist as D
    from shapegen_amd import metrics as M
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env("gloo")
    # the metric kernels need the GPU; the sharding / gathering layer around them is what runs here
    def fake_pair_metrics(a, b, approx=False):
        rows = torch.full((len(a), 3), float("nan"))
        for i, (o, r) in enumerate(zip(a, b)):
            if o.shape[0] and r.shape[0]:
                rows[i] = torch.tensor([float(o.sum() + r.sum()), o.shape[0] * 1.0, r.shape[0] * (2.0 if approx else 1.0)])
        return rows
    M.pair_metrics = fake_pair_metrics
    total = 5
    lo, hi = D.shard_range(total, rank, world)
    orig = [torch.full((i + 1, 3), float(i)) for i in range(total)]
    recon = [torch.full((2, 3), 0.5) if i != 3 else torch.zeros(0, 3) for i in range(total)]       # sample 3: empty cloud
    rows, mean = D.evaluate_sharded(orig[lo:hi], recon[lo:hi], use_approximate_gpu_emd=True)
    assert rows.shape == (total, 3)
    for i in range(total):
        if i == 3:
            assert torch.isnan(rows[i]).all()
        else:
            assert torch.equal(rows[i], torch.tensor([3.0 * i * (i + 1) + 3.0, i + 1.0, 4.0]))
    keep = [i for i in range(total) if i != 3]
    assert torch.allclose(mean, rows[keep].mean(0))
    # dense (B, N, 3) inputs take the same path
    a = torch.arange(total * 6, dtype=torch.float32).reshape(total, 2, 3)
    rows2, _ = D.evaluate_sharded(a[lo:hi], a[lo:hi] * 2)
    assert torch.equal(rows2[:, 0], (a.sum((1, 2)) * 3))
    # shard_context: ranks read disjoint sub-blocks of one global Philox draw
    class _M:
        _shard = None
    from shapegen_amd.diffusion import _DiffusionBase
    m = _M()
    with D.shard_context(m, lo, total):
        off, span = _DiffusionBase._philox_span(m, (hi - lo) * 64 * 3, hi - lo)
    assert (off, span) == (lo * 48, total * 48) and m._shard is None
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"eval_ok{rank}"), "w").write("ok")


def test_two_rank_evaluate_sharded(tmp_path):
    port = _free_port()
    mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "eval_ok0").exists() and (tmp_path / "eval_ok1").exists()


def test_bench_parent_stays_gpu_free_and_reports_failed_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent spawns the ranks before importing torch.  Here (no
    GPU) every rank fails, so the parent must exit non-zero and say which ranks failed."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("the failure path needs a box without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "ranks failed" in p.stderr


@pytest.mark.parametrize("mode,want", [("none", 0), ("exit", 1), ("hang", 124)])
def test_launcher_kills_siblings_of_a_dead_rank_and_enforces_the_timeout(tmp_path, mode, want):
    """`shapegen_amd.launcher.launch_ranks` (bench.py's N > 1 parent): a rank that dies after the rendezvous leaves its peer
    in a barrier; the parent must notice the exit code, terminate the peer, return non-zero well inside the timeout and
    quote the dead rank's stderr.  A rank that hangs is cut by the overall timeout (exit 124)."""
    import sys
    import time
    from shapegen_amd import launcher
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FAULT_MODE=mode, FAULT_RANK="1", PYTHONPATH=root)
    t0 = time.monotonic()
    code, out0, report = launcher.launch_ranks([sys.executable, os.path.join(root, "tests", "launch_worker.py")], 2,
                                               timeout_s=120.0 if mode != "hang" else 25.0, log_dir=str(tmp_path), env=env)
    took = time.monotonic() - t0
    assert code == want, report
    if mode == "none":
        assert '{"ok": true}' in out0 and report == ""
    elif mode == "exit":
        assert took < 60 and "(1, 3)" in report and "simulated failure" in report and "siblings terminated" in report
    else:
        assert 25 <= took < 60 and "timeout" in report
    assert (tmp_path / "rank1.err").exists()              # per-rank logs, not DEVNULL


def test_launcher_gives_every_rank_a_thread_budget_and_a_disjoint_cpu_slice(tmp_path):
    """World 8 over gloo: every rank reports the environment and the affinity mask it actually runs with.  The slices are disjoint, equal in size,
    inside the launcher's own usable CPUs, applied before the rank's first instruction (`sched_getaffinity` in the rank == PCD_RANK_CPUS), and the
    thread budget equals the slice."""
    import json
    import sys
    from shapegen_amd import launcher
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    env.update(FAULT_MODE="none", PYTHONPATH=root, PCD_ENV_DUMP_DIR=str(tmp_path))
    world = 8
    code, out0, report = launcher.launch_ranks([sys.executable, os.path.join(root, "tests", "launch_worker.py")], world,
                                               timeout_s=240.0, log_dir=str(tmp_path), env=env)
    assert code == 0, report
    usable = launcher.usable_cpus()
    per = len(usable) // world
    seen = []
    for r in range(world):
        got = json.load(open(tmp_path / f"env{r}.json"))
        want = usable[r * per:(r + 1) * per] if per >= 1 else usable
        assert got["PCD_RANK_CPUS"] == launcher.cpu_list(want) and got["affinity"] == want, (r, got)
        assert got["OMP_NUM_THREADS"] == got["MKL_NUM_THREADS"] == str(len(want))
        assert got["LOCAL_WORLD_SIZE"] == str(world) and got["LOCAL_RANK"] == str(r) and got["MASTER_ADDR"] == "127.0.0.1"
        assert got["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        seen += want
    if per >= 1:
        assert len(set(seen)) == len(seen) == per * world          # disjoint
    # slices by hand: 20 CPUs over 8 ranks -> 2 each, 4 left over; fewer CPUs than ranks -> nothing to partition
    assert [launcher.rank_cpus(r, 8, range(20)) for r in (0, 7)] == [[0, 1], [14, 15]]
    assert launcher.rank_cpus(5, 8, [3, 4]) == [3, 4]


def test_rank_started_by_torchrun_pins_itself(tmp_path):
    """The driver starts the N > 1 bench through `python -m torch.distributed.run`: bench.py's ranks then call `launcher.apply_rank_affinity()` before
    importing torch.  A child with LOCAL_RANK=1, LOCAL_WORLD_SIZE=2 must end up on the second half of its CPUs with a matching thread budget; a one-rank
    world and a rank the launcher already pinned are left alone."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, os, sys; sys.path.insert(0, %r); from shapegen_amd import launcher; before = launcher.usable_cpus(); "
            "got = launcher.apply_rank_affinity(); print(json.dumps({'before': before, 'got': got, 'aff': sorted(os.sched_getaffinity(0)), "
            "'omp': os.environ.get('OMP_NUM_THREADS')}))" % root)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "PCD_RANK_CPUS", "OMP_NUM_THREADS")}

    def run(**extra):
        p = subprocess.run([sys.executable, "-c", code], env={**base, **extra}, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        return json.loads(p.stdout.strip().splitlines()[-1])

    r = run(LOCAL_RANK="1", LOCAL_WORLD_SIZE="2", WORLD_SIZE="2", RANK="1")
    half = len(r["before"]) // 2
    if half >= 1:
        assert r["aff"] == r["before"][half:2 * half] and r["omp"] == str(half) and r["got"]["PCD_RANK_CPUS"]
    assert run(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")["got"] is None
    already = run(LOCAL_RANK="1", LOCAL_WORLD_SIZE="2", WORLD_SIZE="2", RANK="1", PCD_RANK_CPUS="0")
    assert already["got"] is None and already["aff"] == already["before"]


def test_launcher_told_to_stop_takes_its_ranks_with_it(tmp_path):
    """ranks are session leaders, so a SIGTERM to the launcher (a scheduler, `timeout`, a CI driver) no longer reaches them by itself.
    The launcher's handler must run `_stop`: after SIGTERM to a launcher whose rank 1 hangs, no rank is left alive."""
    import json
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); from shapegen_amd import launcher; "
            "launcher.launch_ranks([sys.executable, %r], 2, timeout_s=300.0, log_dir=%r)"
            % (root, os.path.join(root, "tests", "launch_worker.py"), str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FAULT_MODE="hang", FAULT_RANK="1", PYTHONPATH=root, PCD_ENV_DUMP_DIR=str(tmp_path), PCD_DUMP_EARLY="1")
    parent = subprocess.Popen([sys.executable, "-c", code], env=env)
    try:
        t_end = time.monotonic() + 120
        while not all((tmp_path / f"pid{r}").exists() for r in range(2)):
            assert time.monotonic() < t_end and parent.poll() is None, "ranks did not come up"
            time.sleep(0.1)
        pids = [int(open(tmp_path / f"pid{r}").read()) for r in range(2)]
        parent.send_signal(signal.SIGTERM)
        assert parent.wait(timeout=60) == 128 + signal.SIGTERM
        t_end = time.monotonic() + 20
        alive = pids
        while alive and time.monotonic() < t_end:
            alive = [p for p in alive if os.path.exists(f"/proc/{p}") and open(f"/proc/{p}/stat").read().split(")")[-1].split()[0] != "Z"]
            time.sleep(0.1)
        assert not alive, f"ranks left behind: {alive}"
    finally:
        if parent.poll() is None:
            parent.kill()


def test_launcher_does_not_signal_a_group_it_has_reaped():
    """`_stop` used to `killpg` the pid of a rank it had already reaped -- a pid the kernel may have given to an unrelated session leader.
    A finished rank stays an unreaped zombie until `_stop` has swept its group; after the reap `_signal_group` refuses."""
    import subprocess
    import sys
    import time
    from shapegen_amd import launcher
    p = subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(7)"], start_new_session=True)
    t_end = time.monotonic() + 30
    while launcher._exit_code(p) is None and time.monotonic() < t_end:
        time.sleep(0.02)
    assert launcher._exit_code(p) == 7 and p.returncode is None           # known to have exited, NOT reaped: the pid is still ours
    assert os.path.exists(f"/proc/{p.pid}")
    launcher._stop([p])
    assert p.returncode == 7
    called = []
    real = os.killpg
    os.killpg = lambda *a: called.append(a)
    try:
        launcher._signal_group(p, 9)
    finally:
        os.killpg = real
    assert called == []


class _ToyModel(torch.nn.Module):
    """Stands in for a HIP-trained module in `training.fit`: the control flow under test (how batches are dealt to ranks,
    which collectives pair up, what is broadcast) does not depend on the kernels."""

    def __init__(self, rank):
        super().__init__()
        self.w = torch.nn.Parameter(torch.full((3,), float(rank + 1)))     # ranks start DIFFERENT: fit must broadcast rank 0
        self.register_buffer("stat", torch.full((2,), float(10 * (rank + 1))))
        self.hparams = {}
        self.steps, self.seen = 0, []
        outer = self

        class _Opt:
            lr = 1e-3

            def step(self_inner):
                g = outer.w.detach().clone()
                dist.all_reduce(g)                                           # one collective per optimizer step, like the trainers
                outer.steps += 1

        class _Sched:
            def step(self_inner, metric):
                outer.sched_metric = metric

        self._opt, self._sched = _Opt(), _Sched()

    @property
    def device(self):
        return torch.device("cpu")

    def configure_optimizers(self):
        return {"optimizer": self._opt, "lr_scheduler": {"scheduler": self._sched, "monitor": "val_loss"}}

    def training_step(self, batch, i):
        self.seen.append(i)
        return batch.mean()

    def validation_step(self, batch, i):
        return batch.mean() + dist.get_rank()                              # ranks disagree: fit must average


class _ToyData:
    def __init__(self, n):
        self.n = n

    def setup(self):
        pass

    def train_dataloader(self):
        return [torch.full((2, 2), float(i)) for i in range(self.n)]

    def val_dataloader(self):
        return [torch.zeros(2, 2)]


def _fit_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shapegen_amd  # noqa: F401
    from shapegen_amd import dist as D
    from shapegen_amd.training import fit
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env("gloo")
    model = _ToyModel(rank)
    torch.manual_seed(1000 + 17 * rank)                                     # ranks enter fit() seeded DIFFERENTLY ...
    hist = fit(model, _ToyData(5), max_epochs=2, log=lambda *_: None)       # 5 batches, 2 ranks: the odd one is dropped
    drew = torch.tensor([float(torch.rand(1))])                             # ... and leave it on ONE shared stream (rank 0's seed, broadcast):
    both = [torch.zeros(1), torch.zeros(1)]                                 # the loaders' RandomSampler seeds itself from it, so the ranks
    dist.all_gather(both, drew)                                             # enumerate one permutation
    assert torch.equal(both[0], both[1]), "ranks left fit() on different global RNG streams"
    assert torch.initial_seed() == 1000
    assert torch.equal(model.w.detach(), torch.ones(3)) and torch.equal(model.stat, torch.full((2,), 10.0))   # rank 0's values
    assert model.steps == 4 and model.seen == ([0, 2, 0, 2] if rank == 0 else [1, 3, 1, 3])
    assert abs(hist[-1][2] - 0.5) < 1e-12 and abs(model.sched_metric - 0.5) < 1e-12      # val_loss = mean over ranks of (0, 1)
    dist.barrier()                                                          # no unmatched collective is left behind
    dist.destroy_process_group()
    open(os.path.join(tmp, f"fit_ok{rank}"), "w").write("ok")


def test_two_rank_fit_with_odd_batch_count(tmp_path):
    """ranks must take the same number of optimizer steps (each step is an all-reduce), start from rank 0's
    parameters / buffers, and feed ONE averaged val_loss to the scheduler."""
    port = _free_port()
    mp.spawn(_fit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "fit_ok0").exists() and (tmp_path / "fit_ok1").exists()
