"""GPU-free parent of an N-rank run on one node (one process per GPU, RCCL rendezvous on 127.0.0.1).

Standard library only: the parent must never import torch.cuda or touch a GPU (on this pool a process that has
initialised the GPU may not exec or be replaced, and the ranks need the devices to themselves).  The reference has no
launcher (single device, `train_point_ddpm.py:80-85`); this is build-side.

`launch_ranks` starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, polls ALL of them, and
  * on the first non-zero exit terminates the siblings (a rank that died at RCCL init would otherwise leave the
    others waiting in the rendezvous until the collective timeout),
  * enforces an overall wall-clock limit,
  * keeps every rank's stdout / stderr in per-rank files and puts their tails into the failure message,
  * relays rank 0's stdout (the one JSON line of bench.py) only when every rank succeeded,
  * stops the ranks when the launcher itself is told to stop (SIGTERM / SIGHUP handlers for the duration of the run; each rank also gets
    `PR_SET_PDEATHSIG`, so that a launcher killed with SIGKILL does not leave ranks holding the GPUs),
  * gives each rank a host-thread budget and a disjoint CPU slice (`rank_cpus`): `OMP_NUM_THREADS` / `MKL_NUM_THREADS` = usable cores / world
    and `sched_setaffinity` applied in the child before it runs any Python (8 ranks x torch's default all-core intra-op pool on one host is
    a weak-scaling risk).  Ranks that `torch.distributed.run` started instead call `apply_rank_affinity()` themselves before importing torch.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence, Tuple


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def usable_cpus() -> List[int]:
    """CPUs this process may run on (affinity mask), capped by the cgroup CPU quota: the first `quota` of them."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cpus = cpus[:max(1, int(int(quota) / int(period)))]
    except (OSError, ValueError):
        pass
    return cpus or [0]


def rank_cpus(rank: int, world: int, cpus: Optional[Sequence[int]] = None) -> List[int]:
    """Rank `rank`'s slice of the usable CPUs: `world` contiguous, disjoint slices of equal size (the remainder stays unused, so no rank is
    wider than another); with fewer CPUs than ranks every rank keeps them all (nothing to partition)."""
    cpus = list(usable_cpus() if cpus is None else cpus)
    per = len(cpus) // world
    if per < 1:
        return cpus
    return cpus[rank * per:(rank + 1) * per]


def cpu_list(cpus: Sequence[int]) -> str:
    return ",".join(str(c) for c in cpus)


def apply_rank_affinity(environ=None) -> Optional[Dict[str, str]]:
    """For a rank this launcher did NOT start (`python -m torch.distributed.run ... bench.py`): pin the calling process to its slice and set its
    thread budget, from LOCAL_RANK / LOCAL_WORLD_SIZE (or WORLD_SIZE).  Call before `import torch` (the OpenMP pool reads the variables once).
    Returns what it set, None in a one-rank world or when the launcher already did it (PCD_RANK_CPUS present)."""
    environ = os.environ if environ is None else environ
    world = int(environ.get("LOCAL_WORLD_SIZE", environ.get("WORLD_SIZE", "1")))
    if world <= 1 or "PCD_RANK_CPUS" in environ:
        return None
    rank = int(environ.get("LOCAL_RANK", environ.get("RANK", "0"))) % world
    mine = rank_cpus(rank, world)
    try:
        os.sched_setaffinity(0, mine)
    except (OSError, AttributeError):
        pass
    got = {"PCD_RANK_CPUS": cpu_list(mine), "OMP_NUM_THREADS": str(len(mine)), "MKL_NUM_THREADS": str(len(mine))}
    environ.update(got)
    return got


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None, collective_timeout_s: int = 300,
             cpus: Optional[Sequence[int]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    mine = rank_cpus(rank, world, cpus)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               # host-thread budget and CPU slice of this rank (the slice itself is applied by `_child_setup` before the rank's exec)
               OMP_NUM_THREADS=str(len(mine)), MKL_NUM_THREADS=str(len(mine)), PCD_RANK_CPUS=cpu_list(mine),
               HSA_ENABLE_IPC_MODE_LEGACY="0",
               # a rank stuck in a collective whose peer died raises (and exits non-zero) instead of hanging: the
               # watchdog tears the communicator down after the timeout set in init_process_group
               TORCH_NCCL_ASYNC_ERROR_HANDLING="1",
               PCD_COLLECTIVE_TIMEOUT_S=str(collective_timeout_s))
    return env


def _tail(path: str, limit: int = 1500) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            n = f.tell()
            f.seek(max(0, n - limit))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def _exit_code(p: subprocess.Popen) -> Optional[int]:
    """Exit code of rank `p` if it has exited, WITHOUT reaping it (`waitid(..., WNOWAIT)`): while the zombie stays, its pid -- which is also
    the pgid of the session it leads -- cannot be given to an unrelated process, so a later `killpg(p.pid)` can only reach this rank's own
    group.  `Popen.poll()` reaps, and is therefore not used before `_stop` has swept the group."""
    if p.returncode is not None:
        return p.returncode
    try:
        info = os.waitid(os.P_PID, p.pid, os.WEXITED | os.WNOHANG | os.WNOWAIT)
    except ChildProcessError:            # reaped behind our back (should not happen): let Popen report what it knows
        return p.poll()
    if info is None:
        return None
    return info.si_status if info.si_code == os.CLD_EXITED else -info.si_status


def _signal_group(p: subprocess.Popen, sig: int) -> None:
    """Ranks are session leaders (`start_new_session=True`): signal the rank's whole process group, so that a rank's own children
    (DataLoader workers) do not outlive it and keep the GPU busy.  Only called while `p` is unreaped (alive or zombie): see `_exit_code`."""
    if p.returncode is not None:         # reaped: its pid / pgid may belong to somebody else by now
        return
    try:
        os.killpg(p.pid, sig)            # pgid == pid for a session leader
    except (OSError, ProcessLookupError):
        pass


def _stop(procs: Sequence[subprocess.Popen], grace_s: float = 5.0) -> None:
    """SIGTERM to the process groups of the ranks still running, SIGKILL to every rank's group after `grace_s` (or at once for a rank that
    has already exited: it sweeps what the rank may have left behind), and only THEN reap the rank."""
    for p in procs:
        if _exit_code(p) is None:
            _signal_group(p, signal.SIGTERM)
    t_end = time.monotonic() + grace_s
    for p in procs:
        while _exit_code(p) is None and time.monotonic() < t_end:
            time.sleep(0.05)
        _signal_group(p, signal.SIGKILL)      # the leader is alive or an unreaped zombie: the group id is still ours
        p.wait()


_PR_SET_PDEATHSIG = 1


def _child_setup(cpus: Sequence[int], parent_pid: int):
    """Runs in the child between fork and exec (no GPU, no torch there): the rank dies with the launcher (`PR_SET_PDEATHSIG`), and runs on its
    own CPU slice from its first instruction."""
    def setup():
        try:
            import ctypes
            ctypes.CDLL(None, use_errno=True).prctl(_PR_SET_PDEATHSIG, signal.SIGTERM, 0, 0, 0)
            if os.getppid() != parent_pid:       # the launcher died between fork and prctl
                os._exit(1)
        except Exception:
            pass
        try:
            os.sched_setaffinity(0, cpus)
        except (OSError, AttributeError, ValueError):
            pass
    return setup


class _Stopped(SystemExit):
    pass


def launch_ranks(argv: Sequence[str], world: int, timeout_s: float = 600.0, log_dir: Optional[str] = None,
                 env: Optional[Dict[str, str]] = None, poll_s: float = 0.1) -> Tuple[int, str, str]:
    """Run `argv` as `world` ranks.  Returns (exit code, rank 0's stdout, failure report).  Exit code 0 only when every
    rank exited 0 within `timeout_s`; 1 when a rank failed; 124 on the overall timeout."""
    port = free_port()
    log_dir = log_dir or tempfile.mkdtemp(prefix="pcd_ranks_")
    os.makedirs(log_dir, exist_ok=True)
    procs: List[subprocess.Popen] = []
    files = []
    why = ""
    cpus = usable_cpus()

    def on_signal(signum, frame):          # a scheduler / `timeout` / CI driver stopping the launcher: run the `finally` below
        raise _Stopped(128 + signum)

    previous = {}
    for sig in (signal.SIGTERM, signal.SIGHUP):
        try:
            previous[sig] = signal.signal(sig, on_signal)
        except ValueError:                 # not the main thread: the caller owns signal handling
            pass
    try:
        for rank in range(world):
            out = open(os.path.join(log_dir, f"rank{rank}.out"), "wb")
            files.append(out)
            err = open(os.path.join(log_dir, f"rank{rank}.err"), "wb")
            files.append(err)
            procs.append(subprocess.Popen(list(argv), env=rank_env(rank, world, port, env, max(30, int(timeout_s // 2)), cpus),
                                          stdout=out, stderr=err, start_new_session=True,
                                          preexec_fn=_child_setup(rank_cpus(rank, world, cpus), os.getpid())))
        t_end = time.monotonic() + timeout_s
        while True:
            codes = [_exit_code(p) for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                why = f"ranks failed (rank, exit code): {bad}; siblings terminated"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_end:
                why = f"timeout after {timeout_s:.0f} s; still running: {[r for r, c in enumerate(codes) if c is None]}"
                break
            time.sleep(poll_s)
    except OSError as e:                   # a rank could not be started (ENOMEM, bad argv): stop the ones that were
        why = f"could not start rank {len(procs)}: {e}; siblings terminated"
    finally:
        _stop(procs)
        for f in files:
            f.close()
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    out0 = ""
    try:
        out0 = open(os.path.join(log_dir, "rank0.out"), "r", errors="replace").read()
    except OSError:
        pass
    if not why:
        return 0, out0, ""
    report = [why, f"per-rank logs: {log_dir}"]
    for rank, p in enumerate(procs):
        report.append(f"--- rank {rank} (exit {p.returncode}) stderr tail ---\n{_tail(os.path.join(log_dir, f'rank{rank}.err'))}")
    return (124 if why.startswith("timeout") else 1), out0, "\n".join(report)


def main_launch(script: str, args: Sequence[str], world: int, timeout_s: float) -> int:
    """bench.py's parent: relay rank 0's stdout on success, the report on stderr otherwise."""
    log_dir = os.environ.get("PCD_RANK_LOG_DIR") or None
    code, out0, report = launch_ranks([sys.executable, script] + list(args), world, timeout_s, log_dir)
    if code == 0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    else:
        print(f"bench.py: {report}", file=sys.stderr)
    return code
