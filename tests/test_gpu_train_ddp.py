"""Data-parallel training (one all-reduce of the flat gradient buffer per step): two ranks on one GPU, gloo rendezvous."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_training_steps_stay_in_sync(tmp_path):
    out = str(tmp_path / "ddp.npy")
    import socket
    with socket.socket() as sock:                      # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, DDP_OUT=out,
                   PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_train_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
    diff, err = np.load(out)
    assert diff > 0                               # the two ranks really had different gradients before the exchange
    assert err < 5e-6                             # and the first update is AdamW on their MEAN (lr 1e-3: updates are ~1e-3)
