#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: samples/s (+ Mrays/s, % fp32 VALU peak) of the RTIOW final
scene, 1200x800 @ 1000 spp per GPU ("configs[1]"), 1 -> 8 GPUs.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full pass of the hot path: render `spp` samples for every pixel of the image on this rank's GPU
(rt_render_accum_device: memset + path-trace kernel + finalize, inputs resident in HBM) and, for N > 1, the
RCCL sum-reduce of the fp32 accumulators onto rank 0. Scaling is WEAK: every rank renders the full config-1
workload for its own sample-index range (rank r takes samples [r*spp, (r+1)*spp)), so the N-GPU job produces an
N*spp-sample image. value = samples all ranks rendered / max-over-ranks wall time.

Rank 0 at N = 1 also times the CPU baseline: the oracle's f64 + BVH restatement of the reference algorithm with the
reference's own sequential-RNG rejection samplers (oracle/, "port": the Rust reference cannot be built here), on
BASELINE configs[0] (400x225 @ 100 spp) over the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz (packed fp32)
FLOPS_STATIC, FLOPS_MOVING = 17.0, 23.0  # algorithmic flops of one sphere test


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="FinalStatic")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--max-depth", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from raytracing_amd import api
    from raytracing_amd.dist import weak_range

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    # RTX_BENCH_BACKEND=gloo is a REHEARSAL switch for boxes with fewer GPUs than ranks (ranks then share GPUs and
    # the reduce goes through host memory): it exercises this file's multi-rank path, it is not a measurement.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    dist = None
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    ctx = api.Context(local_rank)  # raises if the HIP library or the device is missing: no fallback
    world = api.World(args.scene, seed=1, aspect=args.width / args.height)
    ctx.upload(world.scene)
    arrs = world.arrays()
    n_moving = int((np.abs(arrs["dc"]).sum(1) > 0).sum())
    n_static = int(len(arrs["radius"]) - n_moving)
    flops_per_ray = FLOPS_STATIC * n_static + FLOPS_MOVING * n_moving

    acc = torch.zeros(args.height, args.width, 3, dtype=torch.float32, device=dev)
    first, count = weak_range(args.spp, rank)
    params = api.make_params(args.width, args.height, count, max_depth=args.max_depth, seed=world.render_seed,
                             spp_offset=first)

    def step():
        st = ctx.render_accum_device(world.camera, params, acc.data_ptr())  # synchronous on the ctx stream
        if dist is not None:
            if backend == "nccl":
                dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI
                # the library renders on its own stream: the reduce must have read `acc` before the next step's
                # render overwrites it, and a step is only done when its image is on rank 0
                torch.cuda.synchronize()
            else:
                dist.reduce(acc.cpu(), dst=0, op=dist.ReduceOp.SUM)
        return st

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    samples_per_step = args.width * args.height * args.spp * world_size
    rays = sum(int(s.rays_traced) for s in stats)
    kernel_ms = sum(float(s.kernel_ms) for s in stats)
    launches = sum(int(s.n_launches) for s in stats)
    rays_all = torch.tensor([float(rays)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
    rays_total = float(rays_all.item())

    if rank == 0:
        value = samples_per_step * args.steps / dt
        achieved = rays * flops_per_ray / (kernel_ms * 1e-3) / 1e12  # this rank's kernel, TFLOP/s algorithmic
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "samples/sec + Mrays/s, RTIOW final scene 1200x800, 1->8 GPUs; % fp32 peak",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if backend == "nccl" else f"synthetic (REHEARSAL: backend {backend}, ranks may share a GPU)",
            "config": {
                "workload": f"configs[1]: RTIOW final scene (static, {n_static + n_moving} spheres) "
                            f"{args.width}x{args.height} @ {args.spp} spp per GPU, max_depth {args.max_depth}, "
                            f"scene seed 1; N GPUs render N*{args.spp} spp (spp_offset = rank*{args.spp}) + RCCL reduce",
                "scene": args.scene, "width": args.width, "height": args.height, "spp_per_gpu": args.spp,
                "max_depth": args.max_depth, "n_static": n_static, "n_moving": n_moving,
                "parallelism": f"spp-split x{world_size}",
            },
            "mrays_per_s": rays_total / dt / 1e6,
            "rays_per_sample": rays / (args.width * args.height * args.spp * args.steps),
            "fp32_peak_frac": achieved / FP32_VALU_PEAK_TFLOPS,
            "roofline": {
                "bound": "valu",  # fp32 VALU compute: not MFMA-shaped, not HBM-bound
                "kernel": "render_kernel",
                "achieved": achieved,
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                "traffic": traffic,  # HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json)
                "traffic_frac_of_hbm_peak": (traffic / (kernel_ms / max(launches, 1) * 1e-3) / 8e12) if traffic else None,
                "flops_per_ray": flops_per_ray,
                "kernel_ms_per_launch": kernel_ms / max(launches, 1),
                "launches_per_step": launches / args.steps,
            },
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline():
    """The oracle as the reported CPU baseline (never the thing measured as `value`)."""
    from oracle import binding as orc
    from raytracing_amd import api
    W, H, SPP = 400, 225, 100  # BASELINE.json configs[0]
    cores = min(16, os.cpu_count() or 1)
    w = api.World("FinalStatic", seed=1, aspect=W / H)
    p = api.make_params(W, H, SPP, seed=w.render_seed)
    # configs[0] takes ~0.5 s on 16 cores: repeat it so the sample is ~15-30 core-seconds, report the median pass
    runs = []
    for _ in range(5):
        _, sec, rays = orc.render(w.scene, w.camera, p, mode=orc.MODE_REFERENCE, bvh_first=w.bvh_first, bvh_seed=1,
                                  n_threads=cores, precision=64)
        runs.append((sec, rays))
    sec, rays = sorted(runs)[len(runs) // 2]
    return {
        "value": W * H * SPP / sec,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"all of configs[0]: FinalStatic {W}x{H} @ {SPP} spp, f64, BVH, reference samplers, "
                  f"{cores} threads over rows (median of 5 passes, {sec:.2f} s each)",
        "mrays_per_s": rays / sec / 1e6,
        "seconds": sec,
    }


if __name__ == "__main__":
    main()
