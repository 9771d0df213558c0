#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: samples/s (+ Mrays/s, % fp32 VALU peak) of the RTIOW final
scene, 1200x800 @ 1000 spp per GPU ("configs[1]"), 1 -> 8 GPUs.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full pass of the hot path: render `spp` samples for every pixel of the image on this rank's GPU
(rt_render_accum_device: memset + path-trace kernel + finalize, inputs resident in HBM) and, for N > 1, the
RCCL sum-reduce of the fp32 accumulators onto rank 0 (raytracing_amd.dist.render_and_reduce, the function the gloo
test covers).  value = samples all ranks rendered / max-over-ranks wall time.

--mode weak (default): every rank renders the full configs[1] workload for its own sample-index range (rank r takes
samples [r*spp, (r+1)*spp)), so the N-GPU job produces an N*spp-sample image.
--mode strong4k: BASELINE configs[4], 3840x2160 @ 4096 spp in total, sample-partitioned: rank r renders
partition_spp(4096, N, r); "scaling": "strong".

At N = 1 rank 0 also reports, OUTSIDE the timed region: (a) every other single-GPU BASELINE config in a `configs`
array (Random = configs[2]; PerlinNoise2 / PerlinTurbulence / PerlinMarbled = configs[3]; one rank's share of
configs[4]: 3840x2160 @ 512 spp at spp_offset 512), each with kernel ms, samples/s, Mrays/s and its own roofline;
(b) the CPU baseline: the oracle's f64 + BVH restatement of the reference algorithm with the reference's own
sequential-RNG rejection samplers (oracle/, "port": the Rust reference cannot be built here), on BASELINE configs[0]
(400x225 @ 100 spp) over ALL the host cores this process may run on.
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
# SURVEY.md §8(d) "for C4 additionally report noise flops": the reference's arithmetic per noise() call
# (material.rs:466-521: p*freq 3, fractions 3, three quintic fades 21, corner offsets 3, eight gradient dots 16,
# seven lerps 28, (n+1)/2 2 = 76; value noise has no dots / offsets / rescale: 55) and per octave of an adapter
# (material.rs:192-203: sum += n*amp 2, p *= lac 3, amp *= gain 1; turbulence's |2n-1| 3 more).
FLOPS_PERLIN_CALL, FLOPS_VALUE_CALL, FLOPS_FBM_LAYER, FLOPS_TURB_LAYER, FLOPS_MARBLE = 76.0, 55.0, 6.0, 9.0, 8.0


def scene_model(api, world):
    """Algorithmic flops per ray segment (sphere tests) and per Lambertian hit (texture) of a scene."""
    import numpy as np
    arrs = world.arrays()
    n_moving = int((np.abs(arrs["dc"]).sum(1) > 0).sum())
    n_static = int(len(arrs["radius"]) - n_moving)
    s = world.scene
    tex_flops = 0.0
    for i in range(s.n_textures):  # the noise scenes texture every sphere with the same noise (config.rs:255-402)
        t = s.textures[i]
        if t.kind in (api.RT_TEX_VALUE_NOISE, api.RT_TEX_PERLIN_NOISE):
            call = FLOPS_PERLIN_CALL if t.kind == api.RT_TEX_PERLIN_NOISE else FLOPS_VALUE_CALL
            layers = 1 if t.op == api.RT_NOISE_PLAIN else t.layers
            per_layer = {api.RT_NOISE_PLAIN: 0.0, api.RT_NOISE_FBM: FLOPS_FBM_LAYER,
                         api.RT_NOISE_TURBULENCE: FLOPS_TURB_LAYER}[t.op]
            tex_flops = max(tex_flops, layers * (call + per_layer) + (FLOPS_MARBLE if t.marbled else 0.0))
    return {"n_static": n_static, "n_moving": n_moving,
            "flops_per_ray": FLOPS_STATIC * n_static + FLOPS_MOVING * n_moving, "flops_per_noise_hit": tex_flops}


def layout_of(ctx, model):
    """rt_debug_scan_layout of the uploaded scene; for a library build older than that hook (tools/ab_scenes.sh
    compares such builds) the variant every BASELINE scene ran then."""
    if hasattr(ctx.lib, "rt_debug_scan_layout"):
        return ctx.debug_scan_layout()
    return {"noise": 1 if model["flops_per_noise_hit"] else 0, "tile_buffers": 2, "planar_slots": 0, "scan": 0}


def roofline_of(model, rays, samples, kernel_ms, launches, layout):
    """fp32 VALU roofline of the render kernel: algorithmic flops = sphere tests for every ray segment + (noise
    scenes) the texture arithmetic of every hit (hits = rays - samples: each path ends with exactly one miss)."""
    flops = rays * model["flops_per_ray"] + max(rays - samples, 0) * model["flops_per_noise_hit"]
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    # template arguments as rocprofv3 prints them: <noise textures compiled in, two LDS tile buffers per wave, scan mode
    # (0 general lists, 1 with the planar list, 2 no scan: every sphere on the f64 list)>, from the layout the library
    # chose at upload (rt_debug_scan_layout)
    b = lambda v: "true" if v else "false"
    kernel = "render_kernel<%s, %s, %d>" % (b(layout["noise"]), b(layout["tile_buffers"] == 2), layout.get("scan", 0))
    return {"bound": "valu", "kernel": kernel,
            "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP32_VALU_PEAK_TFLOPS,
            "frac_means": "time-to-solution against a FLAT-LIST scan at fp32 VALU peak: algorithmic flops (every sphere "
                          "tested for every ray segment, as hit.rs:100-110 does: flops_per_ray x rays) / kernel time / peak. "
                          "NOT VALU utilisation: the kernel culls ~2/3 of those tests, so it executes fewer flops than it is "
                          "credited with (this figure can pass 1.0); see executed_frac and valu_* for what is issued",
            "flops_per_ray": model["flops_per_ray"],
            "flops_per_noise_hit": model["flops_per_noise_hit"],
            "kernel_ms_per_launch": kernel_ms / max(launches, 1)}


SHADER_CLOCK_HZ, N_SIMD = 2.4e9, 1024  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs at 2.4 GHz


def executed_work(scene, width, height, spp, kernel_ms_per_launch):
    """What the kernel EXECUTES, next to the algorithmic figure: from the committed PMC pass of this workload
    (profiles/valu_pmc.json, written by tools/summarize_prof.py), None when it was measured for other arguments.
    executed_frac is an UPPER bound of the fp32 rate actually issued: every VALU wave-instruction priced as a packed
    FMA (256 flop) over the measured share of active lanes, against the live kernel time."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "valu_pmc.json")))
    except Exception:
        return None
    for e in t.get("entries", []):
        c = e.get("config", {})
        if (c.get("scene"), c.get("width"), c.get("height"), c.get("spp")) == (scene, width, height, spp):
            insts = e["valu_insts_per_launch"]
            lane = e["thread_cycles_valu"] / (64.0 * e["active_inst_valu"])
            sec = kernel_ms_per_launch * 1e-3
            upper = insts * 256.0 * lane / sec / 1e12
            return {"valu_insts_per_launch": insts,
                    "valu_cycles_per_inst_per_simd": sec * SHADER_CLOCK_HZ * N_SIMD / insts,
                    "valu_lane_util": lane,
                    "executed_tflops_upper_bound": upper,
                    "executed_frac": upper / FP32_VALU_PEAK_TFLOPS,
                    "pmc_source": e.get("source")}
    return None


def measured_traffic(scene, width, height, spp):
    """HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json), or None when that
    measurement was made for another workload (the figure goes stale when the arguments change)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return None
    for e in t.get("entries", []):
        c = e.get("config", {})
        if (c.get("scene"), c.get("width"), c.get("height"), c.get("spp")) == (scene, width, height, spp):
            return e.get("bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("weak", "strong4k"), default="weak")
    ap.add_argument("--scene", default="FinalStatic")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--allow-lib-override", action="store_true",
                    help="A/B tooling: honour RTX_HIP_LIB (otherwise a set RTX_HIP_LIB is an error); the JSON line "
                         "names the file that ran either way")
    args = ap.parse_args()
    strong = args.mode == "strong4k"
    args.width = args.width or (3840 if strong else 1200)
    args.height = args.height or (2160 if strong else 800)
    args.spp = args.spp or (4096 if strong else 1000)

    import numpy as np  # noqa: F401
    import torch
    from raytracing_amd import api
    from raytracing_amd import dist as rdist
    if args.allow_lib_override:
        api.allow_lib_override()

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
    model = scene_model(api, world)

    # every element is overwritten by the library's finalize kernel; the library renders on its own stream, so
    # the allocation must have completed on torch's stream before the first render touches the buffer
    acc = torch.empty(args.height, args.width, 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    if strong:
        first, count = rdist.partition_spp(args.spp, world_size, rank)
        total_spp = args.spp
    else:
        first, count = rdist.weak_range(args.spp, rank)
        total_spp = args.spp * world_size
    if count == 0:
        sys.exit("more ranks than samples per pixel")
    last_stats = []

    def render_sum(first_, count_):
        params = api.make_params(args.width, args.height, count_, max_depth=args.max_depth, seed=world.render_seed,
                                 spp_offset=first_)
        last_stats.append(ctx.render_accum_device(world.camera, params, acc.data_ptr()))  # synchronous
        return acc if backend == "nccl" else acc.cpu()

    def step():
        # render + (N > 1) ONE sum-reduce onto rank 0, host-synchronised: a step is done when its image is there
        rdist.render_and_reduce(render_sum, first, count, dist)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    del last_stats[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stats = list(last_stats)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the headline step's image (outside the timed region): every pixel finite, and the mean in [0, 1] as the
    # reference's encode demands (screen.rs:31-34); acc holds per-pixel SUMS over this rank's `count` samples
    headline_finite = bool(torch.isfinite(acc).all().item())
    # (on rank 0 of an N-rank job the reduce has summed all ranks' samples into acc: total_spp of them)
    headline_in_unit_range = bool(((acc >= 0) & (acc <= float(total_spp if rank == 0 else count) * (1.0 + 1e-6))).all().item())
    samples_per_step = args.width * args.height * total_spp
    rays = sum(int(s.rays_traced) for s in stats)
    samples_rank = sum(int(s.samples) for s in stats)
    kernel_ms = sum(float(s.kernel_ms) for s in stats)
    launches = sum(int(s.n_launches) for s in stats)
    rays_all = torch.tensor([float(rays)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
    rays_total = float(rays_all.item())

    if rank == 0:
        value = samples_per_step * args.steps / dt
        roof = roofline_of(model, rays, samples_rank, kernel_ms, launches, layout_of(ctx, model))  # this rank's kernel
        traffic = measured_traffic(args.scene, args.width, args.height, count)
        roof["traffic"] = traffic  # HBM bytes per launch, PMC passes of THIS workload (profiles/), else null
        roof["traffic_frac_of_hbm_peak"] = (traffic / (roof["kernel_ms_per_launch"] * 1e-3) / 8e12) if traffic else None
        ex = executed_work(args.scene, args.width, args.height, count, roof["kernel_ms_per_launch"])
        roof.update(ex or {"valu_insts_per_launch": None, "valu_cycles_per_inst_per_simd": None, "valu_lane_util": None,
                           "executed_tflops_upper_bound": None, "executed_frac": None, "pmc_source": None})
        roof["launches_per_step"] = launches / args.steps
        if strong:
            workload = (f"configs[4]: RTIOW final scene (static, {model['n_static'] + model['n_moving']} spheres) "
                        f"{args.width}x{args.height} @ {args.spp} spp in total, sample-partitioned over {world_size} GPU(s) "
                        f"({count} spp on rank 0) + RCCL reduce, max_depth {args.max_depth}, scene seed 1")
        else:
            what = (f"configs[1]: RTIOW final scene (static, {model['n_static'] + model['n_moving']} spheres)"
                    if args.scene == "FinalStatic" else
                    f"scene {args.scene} ({model['n_static']} static + {model['n_moving']} moving spheres; NOT the headline workload)")
            workload = (f"{what} {args.width}x{args.height} @ {args.spp} spp per GPU, max_depth {args.max_depth}, "
                        f"scene seed 1; N GPUs render N*{args.spp} spp (spp_offset = rank*{args.spp}) + RCCL reduce")
        out = {
            "metric": "samples/sec + Mrays/s, RTIOW final scene 1200x800, 1->8 GPUs; % fp32 peak",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if backend == "nccl" else f"synthetic (REHEARSAL: backend {backend}, ranks may share a GPU)",
            "config": {
                "workload": workload,
                "scene": args.scene, "width": args.width, "height": args.height, "spp_per_gpu": count,
                "max_depth": args.max_depth, "n_static": model["n_static"], "n_moving": model["n_moving"],
                "parallelism": f"spp-split x{world_size}",
            },
            "mrays_per_s": rays_total / dt / 1e6,
            "rays_per_sample": rays / samples_rank,
            "all_finite": headline_finite, "in_unit_range": headline_in_unit_range,
            "fp32_peak_frac": roof["frac"],  # = roofline.frac: algorithmic (flat-list) flops / time / peak, see roofline.frac_means
            "roofline": roof,
            "library": api.lib_provenance(),  # the file that ran: absolute path, sha256, whether it is the in-tree build
        }
        if world_size == 1 and not args.no_extra_configs:
            del acc
            out["configs"] = extra_configs(api, torch, ctx, dev)
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# The other single-GPU BASELINE configs (outside the timed region of `value`): (tag, scene, W, H, spp, spp_offset)
EXTRA = (
    ("configs[2] motion blur", "Random", 1200, 800, 500, 0),
    ("configs[3] Perlin noise", "PerlinNoise2", 1200, 800, 500, 0),
    ("configs[3] Perlin turbulence", "PerlinTurbulence", 1200, 800, 500, 0),
    ("configs[3] Perlin marble", "PerlinMarbled", 1200, 800, 500, 0),
    ("configs[4] one rank's share (rank 1 of 8)", "FinalStatic", 3840, 2160, 512, 512),
)


def extra_configs(api, torch, ctx, dev, runs=3):
    res = []
    for tag, scene, W, H, spp, off in EXTRA:
        w = api.World(scene, seed=1, aspect=W / H)
        ctx.upload(w.scene)
        model = scene_model(api, w)
        acc = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        p = api.make_params(W, H, spp, seed=w.render_seed, spp_offset=off)
        ctx.render_accum_device(w.camera, p, acc.data_ptr())  # warm-up
        sts = [ctx.render_accum_device(w.camera, p, acc.data_ptr()) for _ in range(runs)]
        st = sorted(sts, key=lambda s: s.kernel_ms)[len(sts) // 2]  # median run
        finite = bool(torch.isfinite(acc).all().item())
        del acc
        roof = roofline_of(model, int(st.rays_traced), int(st.samples), float(st.kernel_ms), int(st.n_launches),
                           layout_of(ctx, model))
        roof["traffic"] = measured_traffic(scene, W, H, spp)
        roof.update(executed_work(scene, W, H, spp, roof["kernel_ms_per_launch"]) or {})
        res.append({
            "config": tag, "scene": scene, "width": W, "height": H, "spp": spp, "spp_offset": off,
            "n_static": model["n_static"], "n_moving": model["n_moving"],
            "kernel_ms": float(st.kernel_ms), "total_ms": float(st.total_ms), "launches": int(st.n_launches),
            "samples_per_s": st.samples / (st.total_ms * 1e-3), "mrays_per_s": st.rays_traced / (st.total_ms * 1e-3) / 1e6,
            "rays_per_sample": st.rays_traced / st.samples, "all_finite": finite, "roofline": roof,
        })
    return res


def host_cores():
    """CPUs this process may run on (the reference's rayon pool would use exactly these), and the machine's count."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:  # a cgroup CPU quota below the affinity count is the real limit
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    return usable, os.cpu_count() or usable


def cpu_baseline():
    """The oracle as the reported CPU baseline (never the thing measured as `value`)."""
    from oracle import binding as orc
    from raytracing_amd import api
    W, H, SPP = 400, 225, 100  # BASELINE.json configs[0]
    cores, nproc = host_cores()
    w = api.World("FinalStatic", seed=1, aspect=W / H)
    p = api.make_params(W, H, SPP, seed=w.render_seed)
    # configs[0] takes ~0.5 s on 16 cores: repeat it so the sample is ~10-30 core-seconds, report the median pass
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
        "host_cpus": nproc,
        "kind": "port",
        "sample": f"all of configs[0]: FinalStatic {W}x{H} @ {SPP} spp, f64, BVH, reference samplers, "
                  f"{cores} threads over rows = every CPU this process may run on (machine reports {nproc}) "
                  f"(median of 5 passes, {sec:.2f} s each)",
        "mrays_per_s": rays / sec / 1e6,
        "seconds": sec,
    }


if __name__ == "__main__":
    main()
