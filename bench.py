"""Headline benchmark: denoising-steps/sec of the point-cloud diffusion sampler hot path.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg5] [--backbone pointnet|attention]

Default (`--config cfg2`, BASELINE.json configs[1]): one "step" = one ancestral denoising step (`sample2`
loop body, reference diffusion.py:241-257) of a whole batch: denoiser forward (HIP kernels) + on-device Philox
noise + fused DDPM update; 2048 points, batch 64 per GPU, fp16 operands / fp32 accumulation, 1000-step
schedule (the K timed steps are the first steps of that schedule).  Inputs are resident in HBM before the timed
region.  With N > 1 every rank runs the same per-GPU batch (weak scaling, no data-path collective: shapes are
independent).

Launching: with N > 1 and no WORLD_SIZE in the environment this process never touches the GPU: it starts N
child processes (one rank per GPU, RCCL rendezvous on 127.0.0.1), relays rank 0's JSON line and exits non-zero if
any rank failed.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are
already there and each process is one rank.

Other workloads (parity-test configurations of BASELINE.json, reported with the same JSON contract):
  --config cfg3   DDIM `sample`, 50 steps, 64 shapes per GPU through `dist.sample_sharded` (clouds all-gathered)
  --config cfg5   latent diffusion: 1000-step latent DDIM, 32 latents per GPU, VAE decode, voxel->points,
                  RCCL all-gather of the ragged clouds, per-sample Chamfer / Sinkhorn-EMD / voxel-BCE rows

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":            dominant kernel (global_feat.3 GEMM 2048->4096 with fused max) timed live with HIP
                         events on the launch stream, against the dense fp16 MFMA peak;
  "roofline_attention":  the set-attention kernel (QK^T / softmax / PV, reference networks.py:61,80-81) at
                         batch 64 x 2048 points, C = 256, 4 heads, timed the same way (N = 1 only);
  "cpu_baseline":        the CPU oracle (PyTorch-CPU fp32 restatement) timed on the host cores on a bounded
                         sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
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
ATT_C, ATT_HEADS = 256, 4
ATT_FLOP_PER_LAUNCH = 4.0 * B_PER_GPU * N_POINTS * N_POINTS * ATT_C      # QK^T + PV (4 B N^2 C)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 x ~4 ms; the GPU reaches steady clocks after ~10 steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=("cfg2", "cfg3", "cfg5"), default="cfg2")
    ap.add_argument("--backbone", choices=("pointnet", "attention"), default="pointnet",
                    help="denoiser behind the cfg2 sampler: UNetPointNetLarge (the reference's wiring) or "
                         "UNetAttentionPointExperimental (the set-attention carrier)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-attention", action="store_true", help="skip the roofline_attention leg")
    ap.add_argument("--graph", action="store_true",
                    help="time the hipGraph replay path in the headline region instead of eager launches (the eager "
                         "default lets the dominant kernel be timed by HIP events inside the timed region; the graph "
                         "number is reported beside it either way)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ launcher (GPU-free)
def launch_children(args) -> int:
    """Parent of an N-rank run.  Runs BEFORE anything imports torch.cuda: starts one child per GPU, waits,
    relays rank 0's stdout, returns non-zero if any rank failed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0 = procs[0].communicate()[0]
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------ helpers
def synth_weights(backbone="pointnet"):
    import numpy as np
    import torch
    from shapegen_amd import specs
    spec = (specs.unet_pointnet_large_spec(prefix="model.") if backbone == "pointnet"
            else specs.unet_attention_spec(prefix="model."))
    gain = 1.3 if backbone == "pointnet" else 1.0
    return {k: torch.from_numpy(np.asarray(v)) for k, v in specs.synth_state_dict(spec, seed=0, gain=gain).items()}


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


def cpu_baseline(sd, sample_batch=2, reps=4):
    """Oracle (port of the reference's PyTorch-CPU path) on a bounded sample: `reps` DDPM steps at
    B=sample_batch, N=2048 after one warm-up step, scaled to the B=64 step by the batch ratio."""
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
                      f"batch) after one warm-up step, scaled x{B_PER_GPU // sample_batch}; torch CPU fp32, {cores} threads"}


class Ranks:
    """Process-group plumbing of one rank."""

    def __init__(self, args):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # PCD_BENCH_SHARE_GPU=1 (tests on a one-GPU box only): every rank uses cuda:0 and the collectives run
        # over gloo, staged through the host; RCCL needs one device per rank
        self.shared = os.environ.get("PCD_BENCH_SHARE_GPU") == "1"
        self.device = torch.device("cuda", 0 if self.shared else self.local)
        torch.cuda.set_device(self.device)
        self.dist = None
        self.backend = None
        if self.world > 1:
            import torch.distributed as dist
            self.backend = "gloo" if self.shared else "nccl"
            kw = {} if self.shared else {"device_id": self.device}
            dist.init_process_group(self.backend, **kw)
            self.dist = dist

    def sync_all(self):
        import torch
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        import torch
        if self.dist is None:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.shared else self.device)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def collective_ranks(self):
        """World size as seen by a collective on the data-path backend (an all-gather of one int per rank)."""
        import torch
        if self.dist is None:
            return 1
        one = torch.ones(1, dtype=torch.int32, device="cpu" if self.shared else self.device)
        got = [torch.zeros_like(one) for _ in range(self.world)]
        self.dist.all_gather(got, one)
        return int(sum(int(g.item()) for g in got))

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def attention_roofline(device, launches=100):
    """HIP events (torch's current stream is the stream the kernel is launched on) around
    pcd_set_attention_f16 at B=64, N=2048, C=256, 4 heads."""
    import torch
    from shapegen_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(7)
    qkv = torch.randn(B_PER_GPU * N_POINTS, 3 * ATT_C, generator=g).to(device, torch.float16)     # unit-variance q, k, v
    out = torch.empty(B_PER_GPU * N_POINTS, ATT_C, dtype=torch.float16, device=device)

    def launch():
        _lib.check(lib.pcd_set_attention_f16(qkv.data_ptr(), B_PER_GPU, N_POINTS, ATT_C, ATT_HEADS, out.data_ptr(), 0, 0,
                                             _lib.stream_ptr()), "set_attention")
    # warm-up: the chip takes ~100 launches (27 ms) from idle to its running clocks (tools/bench_attn_sustained.py: the first
    # 100-launch chunk measures 10 % below the following nine); the timed train starts after that ramp
    for _ in range(200):
        launch()
    # one event pair around the whole train of launches (an event record between kernels costs tens of microseconds of
    # its own, comparable to the kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    if not torch.isfinite(out.float()).all():
        raise SystemExit("set attention produced non-finite values")
    achieved = ATT_FLOP_PER_LAUNCH / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "set_attention_sp_kernel (QK^T, softmax, PV; d_head 64; software-pipelined, 2 query blocks per wave)",
            "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": None,
            "avg_launch_ms": ms, "launches_timed": launches, "flop_per_launch": ATT_FLOP_PER_LAUNCH,
            "shape": {"batch": B_PER_GPU, "points": N_POINTS, "channels": ATT_C, "heads": ATT_HEADS}}


def measured_traffic():
    """HBM/fabric bytes per launch of the dominant kernel from the committed PMC recipe (tools/pmc_gf3.sh writes
    profiles/gf3_pmc_latest.json with the git hash it was taken at); None when absent."""
    pmc = os.path.join(ROOT, "profiles", "gf3_pmc_latest.json")
    try:
        rec = json.load(open(pmc))
        return rec.get("hbm_bytes_per_launch"), rec.get("git_head")
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------ cfg2 (headline)
def run_cfg2(args, R: Ranks):
    import torch
    from shapegen_amd import _lib
    from shapegen_amd.diffusion import PointCloudDiffusion, Stepper

    sd = synth_weights(args.backbone)
    kw = {} if args.backbone == "pointnet" else {"backbone": "attention"}
    model = PointCloudDiffusion(num_points=N_POINTS, **kw)
    model.load_state_dict(sd, strict=True)
    model = model.to(R.device).eval()
    total = args.warmup + args.steps
    if total + 1 > SCHEDULE_STEPS:
        raise SystemExit("warmup+steps must be < 1000")
    tab = model.ddpm_table(SCHEDULE_STEPS, B_PER_GPU)
    bias = model.model.time_bias(tab.t)
    lib = _lib.load()
    pointnet = args.backbone == "pointnet"
    handle = model.model._handle if pointnet else None

    def new_stepper(seed):
        torch.manual_seed(seed)
        x = model._randn_like(torch.empty(B_PER_GPU, N_POINTS, 3, device=model.device))   # x_T resident in HBM
        # the product's own step object (diffusion.Stepper, what sample2() drives): device-side step select +
        # denoiser forward + Philox noise + fused DDPM update, state updated in place
        return x, Stepper(model, x, tab, bias, model._forward_fn(), "ddpm")

    def leg(use_graph, seed):
        x, stp = new_stepper(seed)
        for k in range(args.warmup):
            stp.step(k, True)
        if use_graph:
            stp.capture()
        if pointnet:
            _lib.check(lib.pcd_unet_profile(handle, 0 if use_graph else 1))
        R.sync_all()
        t0 = time.perf_counter()
        for k in range(args.warmup, total):
            if use_graph:
                stp.replay()
            else:
                stp.step(k, True)
        R.sync_all()
        elapsed = time.perf_counter() - t0
        tot_ms, launches = C.c_double(0), C.c_int(0)
        if pointnet:
            _lib.check(lib.pcd_unet_profile_read(handle, C.byref(tot_ms), C.byref(launches)))
            _lib.check(lib.pcd_unet_profile(handle, 0))
        if not torch.isfinite(x).all():
            raise SystemExit("non-finite state after the timed steps")
        return R.max_over_ranks(elapsed), tot_ms.value, launches.value

    # the set-attention kernel's own roofline first, on an idle chip (100 launches = 27 ms; after the ~1 s of the two legs below
    # the chip is power-throttled, which is a statement about the GEMM run that heated it, not about this kernel)
    att = attention_roofline(R.device) if (R.world == 1 and not args.no_attention) else None
    head_graph = args.graph
    elapsed, tot_ms, launches = leg(head_graph, 24 + R.rank)
    other_elapsed, o_ms, o_launches = leg(not head_graph, 24 + R.rank)      # the other launch mode, same K steps
    if head_graph:           # HIP events cannot be recorded inside a captured step: the eager leg carries them
        tot_ms, launches = o_ms, o_launches
    rccl_ranks = R.collective_ranks()
    if R.rank != 0:
        return None
    world = R.world
    name = "UNetPointNetLarge" if pointnet else "UNetAttentionPointExperimental"
    out = {
        "metric": "denoising-steps/sec (whole node), 2048-pt DDPM, batch 64",
        "value": world * args.steps / elapsed,
        "unit": "denoising-steps/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: point-cloud DDPM (sample2), 2048 points, batch 64 per GPU, "
                               f"1000-step cosine schedule, fp16 operands/fp32 accumulate, random-init synthetic weights, "
                               f"denoiser {name}",
                   "batch_per_gpu": B_PER_GPU, "points": N_POINTS, "sampler": "ddpm/sample2", "backbone": args.backbone,
                   "launch": "hipGraph replay" if head_graph else "eager",
                   ("eager" if head_graph else "graph_replay") + "_steps_per_sec": world * args.steps / other_elapsed,
                   "point_steps_per_sec": world * args.steps * B_PER_GPU * N_POINTS / elapsed,
                   "rccl_ranks": rccl_ranks, "collective_backend": R.backend},
    }
    if pointnet:
        gf3_ms = tot_ms / max(launches, 1)
        achieved = GF3_FLOP_PER_LAUNCH / (gf3_ms * 1e-3) / 1e12
        traffic, traffic_head = measured_traffic()
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_f16_kernel<256,256,2,4,2,COLMAX> persistent (global_feat.3 2048->4096 + max over N)",
                           "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_measured_at": traffic_head,
                           "avg_launch_ms": gf3_ms, "launches_timed": launches,
                           "flop_per_launch": GF3_FLOP_PER_LAUNCH}
    if att is not None:
        att["measured"] = "before the timed region, after 200 warm-up launches"
        out["roofline_attention"] = att
        if not pointnet:
            out["roofline"] = att
    if world == 1 and not args.no_cpu_baseline and pointnet:
        out["cpu_baseline"] = cpu_baseline(sd)
    return out


# ------------------------------------------------------------------------------------------ cfg3
def run_cfg3(args, R: Ranks):
    """BASELINE configs[2]: DDIM `sample`, 2048 points, 50 steps, 64 shapes per GPU (512 on 8 GPUs), batch-sharded;
    the only collective is the all-gather of the output clouds after the loop."""
    import torch
    from shapegen_amd import dist as D
    from shapegen_amd.diffusion import PointCloudDiffusion
    T = 50
    model = PointCloudDiffusion(num_points=N_POINTS)
    model.load_state_dict(synth_weights(), strict=True)
    model = model.to(R.device).eval()
    torch.manual_seed(24)
    runs = max(1, args.steps // T)
    gb = B_PER_GPU * R.world
    for _ in range(max(1, args.warmup // T)):
        D.sample_sharded(model, gb, N_POINTS, T)
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(runs):
        clouds = D.sample_sharded(model, gb, N_POINTS, T)
    R.sync_all()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    ranks = R.collective_ranks()
    assert tuple(clouds.shape) == (gb, N_POINTS, 3) and torch.isfinite(clouds).all()
    if R.rank != 0:
        return None
    steps = runs * T
    return {"metric": "denoising-steps/sec (whole node), 2048-pt DDIM 50 steps, batch 64 per GPU", "value": R.world * steps / elapsed,
            "unit": "denoising-steps/sec", "n_gpus": R.world, "steps": steps, "warmup": max(1, args.warmup // T) * T,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: point-cloud DDIM (sample), 2048 points, 50 steps, batch 64 per GPU "
                                   "batch-sharded, output clouds all-gathered; whole sampler calls timed (tables, graph capture, gather included)",
                       "global_batch": gb, "runs": runs, "rccl_ranks": ranks, "collective_backend": R.backend}}


# ------------------------------------------------------------------------------------------ cfg5
def run_cfg5(args, R: Ranks):
    """BASELINE configs[4]: latent diffusion, 1000 latent DDIM steps on 32 latents per GPU, VAE decode, voxel->points,
    all-gather of the ragged clouds, per-sample Chamfer / Sinkhorn EMD / voxel BCE rows against the input grids' clouds."""
    import numpy as np
    import torch
    from shapegen_amd import dist as D
    from shapegen_amd import specs
    from shapegen_amd.diffusion import LatentDiffusion
    from shapegen_amd.utils import voxel_tensor_to_point_clouds
    from shapegen_amd.vae import VAE3DLarge
    B, T = 32, SCHEDULE_STEPS
    sd = specs.synth_state_dict(specs.latent_unet_spec(prefix="model."), seed=0, gain=1.3)
    sd.update(specs.synth_state_dict(specs.vae3d_large_spec(prefix="vae."), seed=0, gain=1.3))
    m = LatentDiffusion(VAE3DLarge())
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(R.device).eval()
    torch.manual_seed(24)
    gb = B * R.world
    lo, hi = D.shard_range(gb, R.rank, R.world)
    g = torch.Generator().manual_seed(24)
    vox = (torch.rand(gb, 1, 32, 32, 32, generator=g) > 0.9).float()[lo:hi].to(R.device)
    orig = voxel_tensor_to_point_clouds(vox, 0.5)

    def once():
        R.sync_all()
        t0 = time.perf_counter()
        with D.shard_context(m, lo, gb):
            clouds = m.sample(num_samples=hi - lo, num_steps=T)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        allc = D.all_gather_clouds(clouds)
        rows, mean = D.evaluate_sharded(orig, clouds, use_approximate_gpu_emd=True)
        R.sync_all()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, allc, rows

    once()
    loop_s, eval_s, allc, rows = once()
    loop_s, eval_s = R.max_over_ranks(loop_s), R.max_over_ranks(eval_s)
    ranks = R.collective_ranks()
    assert len(allc) == gb and rows.shape == (gb, 3)
    if R.rank != 0:
        return None
    return {"metric": "denoising-steps/sec (whole node), latent diffusion 1000 steps, batch 32 per GPU", "value": R.world * T / loop_s,
            "unit": "denoising-steps/sec", "n_gpus": R.world, "steps": T, "warmup": T, "ms_per_step": loop_s / T * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: latent diffusion, 1000 DDIM steps over (32, 256) latents per GPU + VAE3DLarge decode "
                                   "+ voxel->points (timed as the loop), then all-gather of ragged clouds + per-sample Chamfer/Sinkhorn-EMD/voxel-BCE",
                       "global_batch": gb, "eval_seconds": eval_s, "mean_metrics": [float(v) for v in rows.nanmean(dim=0)],
                       "rccl_ranks": ranks, "collective_backend": R.backend}}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args))           # the parent stays GPU-free

    import torch
    import shapegen_amd  # noqa: F401
    torch.set_grad_enabled(False)
    R = Ranks(args)
    out = {"cfg2": run_cfg2, "cfg3": run_cfg3, "cfg5": run_cfg5}[args.config](args, R)
    if R.rank == 0:
        print(json.dumps(out), flush=True)
    R.finish()


if __name__ == "__main__":
    main()
