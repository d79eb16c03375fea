"""GPU-free parent of an N-rank run on one node (one process per GPU, RCCL rendezvous on 127.0.0.1).

Standard library only: the parent must never import torch.cuda or touch a GPU (on this pool a process that has
initialised the GPU may not exec or be replaced, and the ranks need the devices to themselves).  The reference has no
launcher (single device, `train_point_ddpm.py:80-85`); this is build-side.

`launch_ranks` starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, polls ALL of them, and
  * on the first non-zero exit terminates the siblings (a rank that died at RCCL init would otherwise leave the
    others waiting in the rendezvous until the collective timeout),
  * enforces an overall wall-clock limit,
  * keeps every rank's stdout / stderr in per-rank files and puts their tails into the failure message,
  * relays rank 0's stdout (the one JSON line of bench.py) only when every rank succeeded.
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


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None, collective_timeout_s: int = 300) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
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


def _signal_group(p: subprocess.Popen, sig: int) -> None:
    """Ranks are session leaders (`start_new_session=True`): signal the rank's whole process group, so that a rank's own
    children (DataLoader workers) do not outlive it and keep the GPU busy.  Only groups this launcher created are touched."""
    try:
        os.killpg(p.pid, sig)            # pgid == pid for a session leader
    except (OSError, ProcessLookupError):
        try:
            p.send_signal(sig)
        except OSError:
            pass


def _stop(procs: Sequence[subprocess.Popen], grace_s: float = 5.0) -> None:
    """SIGTERM to the process groups of the exact PIDs we started, then SIGKILL to whatever is still alive after `grace_s`."""
    for p in procs:
        if p.poll() is None:
            _signal_group(p, signal.SIGTERM)
    t_end = time.monotonic() + grace_s
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            _signal_group(p, signal.SIGKILL)
            p.wait()
        else:
            _signal_group(p, signal.SIGKILL)      # the rank is gone; sweep what it may have left in its group


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
    try:
        for rank in range(world):
            out = open(os.path.join(log_dir, f"rank{rank}.out"), "wb")
            files.append(out)
            err = open(os.path.join(log_dir, f"rank{rank}.err"), "wb")
            files.append(err)
            procs.append(subprocess.Popen(list(argv), env=rank_env(rank, world, port, env, max(30, int(timeout_s // 2))),
                                          stdout=out, stderr=err, start_new_session=True))
        t_end = time.monotonic() + timeout_s
        while True:
            codes = [p.poll() for p in procs]
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
