"""Headline benchmark: denoising-steps/sec of the point-cloud DDPM sampler hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one ancestral denoising step (`sample2` loop body, reference diffusion.py:241-257)
of a whole batch: UNetPointNetLarge forward (HIP kernels) + on-device Philox noise + fused
DDPM update.  Workload = BASELINE.json configs[1]: 2048 points, batch 64 per GPU, fp16
operands / fp32 accumulation, 1000-step schedule (the K timed steps are the first steps of
that schedule).  Inputs are resident in HBM before the timed region.  With N > 1 every rank
runs the same per-GPU batch (weak scaling, no data-path collective: shapes are independent).

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":     dominant kernel (global_feat.3 GEMM 2048->4096 with fused max) timed live with
                  HIP events on the launch stream, against the dense fp16 MFMA peak;
  "cpu_baseline": the CPU oracle (PyTorch-CPU fp32 restatement) timed on the host cores on a
                  bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_PER_GPU = 64
N_POINTS = 2048
SCHEDULE_STEPS = 1000
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0           # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16
GF3_FLOP_PER_LAUNCH = 2.0 * B_PER_GPU * N_POINTS * 2048 * 4096   # algorithmic FLOP of the dominant GEMM


def synth_weights():
    import numpy as np
    import torch
    from shapegen_amd import specs
    spec = specs.unet_pointnet_large_spec(prefix="model.")
    return {k: torch.from_numpy(np.asarray(v)) for k, v in specs.synth_state_dict(spec, seed=0, gain=1.3).items()}


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sd, sample_batch=2, reps=2):
    """Oracle (port of the reference's PyTorch-CPU path) on a bounded sample: `reps` DDPM steps at
    B=sample_batch, N=2048 after one warm-up, scaled to the B=64 step by the batch ratio."""
    import torch
    from oracle import torch_oracle as O   # checker/baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(24)
    x = torch.randn(sample_batch, N_POINTS, 3, generator=g)
    model = lambda xx, tt: O.unet_pointnet_large(sd, "model.", xx, tt)
    z = [torch.randn(sample_batch, N_POINTS, 3, generator=g) for _ in range(reps + 1)]

    def one(i, xx):
        t = torch.ones(sample_batch) * (SCHEDULE_STEPS - 1 - i) / SCHEDULE_STEPS
        n, s = O.offset_cosine_schedule(t)
        eps = model(xx, t)
        x0 = O.remove_noise(xx, eps, n, s)
        tp = torch.ones(sample_batch) * (SCHEDULE_STEPS - 2 - i) / SCHEDULE_STEPS
        npv, sp = O.offset_cosine_schedule(tp)
        return sp.view(-1, 1, 1) * x0 + (torch.sqrt(npv / n) * n).view(-1, 1, 1) * z[i]

    with torch.no_grad():
        x = one(0, x)
        t0 = time.perf_counter()
        for i in range(1, reps + 1):
            x = one(i, x)
        dt = (time.perf_counter() - t0) / reps
    per_full_step = dt * (B_PER_GPU / sample_batch)
    return {"value": 1.0 / per_full_step, "unit": "denoising-steps/sec", "cores": cores, "kind": "port",
            "sample": f"{reps} DDPM steps at B={sample_batch}, N={N_POINTS} (1/{B_PER_GPU // sample_batch} of the "
                      f"batch), scaled x{B_PER_GPU // sample_batch}; torch CPU fp32, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 50 x ~4 ms; the GPU reaches steady clocks after ~10 steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured HIP graph per step (what sample2() does for long runs); the default launches "
                         "eagerly so that the dominant kernel can be timed by HIP events inside the timed region")
    args = ap.parse_args()

    import torch
    import shapegen_amd  # noqa: F401
    from shapegen_amd import _lib
    from shapegen_amd.diffusion import PointCloudDiffusion

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    torch.set_grad_enabled(False)
    sd = synth_weights()
    model = PointCloudDiffusion(num_points=N_POINTS)
    model.load_state_dict(sd, strict=True)
    model = model.to(torch.device("cuda", local_rank)).eval()

    total = args.warmup + args.steps
    if 2 * total + 1 > SCHEDULE_STEPS:
        raise SystemExit("warmup+steps must be < 1000")
    from shapegen_amd.diffusion import Stepper
    tab = model.ddpm_table(SCHEDULE_STEPS, B_PER_GPU)
    torch.manual_seed(24 + rank)
    x = model._randn_like(torch.empty(B_PER_GPU, N_POINTS, 3, device=model.device))   # x_T resident in HBM
    lib = _lib.load()
    # the product's own step object (diffusion.Stepper, what sample2() drives): device-side step select +
    # UNet forward + Philox noise + fused DDPM update, state updated in place
    stp = Stepper(model, x, tab, model.model.time_bias(tab.t), model._forward_fn(), "ddpm")

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    use_graph = args.graph
    for k in range(args.warmup):
        stp.step(k, True)
    if use_graph:
        stp.capture()
    handle = model.model._handle
    _lib.check(lib.pcd_unet_profile(handle, 0 if use_graph else 1))
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        if use_graph:
            stp.replay()
        else:
            stp.step(k, True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_graph:
        # HIP events cannot be recorded inside the captured step: time the dominant kernel over the same
        # number of eager steps right after the timed region (same stream, same data, same clocks)
        _lib.check(lib.pcd_unet_profile(handle, 1))
        for k in range(total, total + args.steps):
            stp.step(k, True)
        torch.cuda.synchronize()
    tot_ms, launches = C.c_double(0), C.c_int(0)
    _lib.check(lib.pcd_unet_profile_read(handle, C.byref(tot_ms), C.byref(launches)))
    _lib.check(lib.pcd_unet_profile(handle, 0))
    if not torch.isfinite(x).all():
        raise SystemExit("non-finite state after the timed steps")

    if dist is not None:
        tt = torch.tensor([elapsed], device=model.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        gf3_ms = tot_ms.value / max(launches.value, 1)
        achieved = GF3_FLOP_PER_LAUNCH / (gf3_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "gf3_pmc_latest.json")
        if os.path.isfile(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "denoising-steps/sec (whole node), 2048-pt DDPM, batch 64",
            "value": world * args.steps / elapsed,
            "unit": "denoising-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: point-cloud DDPM (sample2), 2048 points, batch 64 per GPU, "
                                   "1000-step cosine schedule, fp16 operands/fp32 accumulate, random-init synthetic weights",
                       "batch_per_gpu": B_PER_GPU, "points": N_POINTS, "sampler": "ddpm/sample2", "launch": "hipGraph replay" if args.graph else "eager",
                       "point_steps_per_sec": world * args.steps * B_PER_GPU * N_POINTS / elapsed},
            "roofline": {"bound": "mfma", "kernel": "gemm_f16_kernel<256,256,2,4,2,COLMAX> persistent (global_feat.3 2048->4096 + max over N)",
                         "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": gf3_ms, "launches_timed": launches.value,
                         "flop_per_launch": GF3_FLOP_PER_LAUNCH},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
