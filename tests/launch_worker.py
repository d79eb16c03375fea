"""Rank of a launcher rehearsal (tests/test_dist_gloo.py): gloo rendezvous, one barrier, then rank FAULT_RANK dies
(FAULT_MODE=exit), hangs (FAULT_MODE=hang) or nobody does; the others wait in a second barrier like ranks whose peer
died at RCCL init would."""
import datetime
import os
import sys
import time

import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world,
                        timeout=datetime.timedelta(seconds=int(os.environ.get("PCD_COLLECTIVE_TIMEOUT_S", "300"))))
dist.barrier()
if os.environ.get("PCD_DUMP_EARLY"):
    with open(os.path.join(os.environ["PCD_ENV_DUMP_DIR"], f"pid{rank}.tmp"), "w") as f:
        f.write(str(os.getpid()))
    os.replace(os.path.join(os.environ["PCD_ENV_DUMP_DIR"], f"pid{rank}.tmp"), os.path.join(os.environ["PCD_ENV_DUMP_DIR"], f"pid{rank}"))
mode, bad = os.environ.get("FAULT_MODE", "none"), int(os.environ.get("FAULT_RANK", "-1"))
if rank == bad and mode == "exit":
    print(f"rank {rank}: simulated failure", file=sys.stderr, flush=True)
    os._exit(3)
if rank == bad and mode == "hang":
    time.sleep(3600)
dist.barrier()
if os.environ.get("PCD_ENV_DUMP_DIR"):            # what the launcher gave this rank (thread budget, CPU slice), for the test to compare
    import json
    with open(os.path.join(os.environ["PCD_ENV_DUMP_DIR"], f"env{rank}.json"), "w") as f:
        json.dump({"affinity": sorted(os.sched_getaffinity(0)), "pid": os.getpid(),
                   **{k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "PCD_RANK_CPUS", "LOCAL_WORLD_SIZE",
                                                     "LOCAL_RANK", "MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY")}}, f)
if rank == 0:
    print('{"ok": true}', flush=True)
dist.destroy_process_group()
