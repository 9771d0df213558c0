"""world_size-2 gloo test of the multi-GPU partition logic on CPU: spp split with spp_offset + one sum-reduce
onto rank 0 equals the single-rank render. The per-rank renderer here is the CPU oracle (tests may use it); on
the GPU box bench.py plugs rt_render_accum_device into the same raytracing_amd.dist functions over RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_spp():
    from raytracing_amd.dist import partition_spp, weak_range
    for total in (0, 1, 7, 8, 1000, 4096):
        for world in (1, 2, 3, 8):
            parts = [partition_spp(total, world, r) for r in range(world)]
            assert sum(c for _, c in parts) == total
            nxt = 0
            for first, count in parts:
                assert first == nxt and count >= 0
                nxt += count
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert weak_range(1000, 3) == (3000, 1000)
    with pytest.raises(ValueError):
        partition_spp(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import binding as orc
    from raytracing_amd import api
    from raytracing_amd.dist import partition_spp, render_and_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, SPP = 48, 28, 9  # odd spp: ranks get 5 and 4
    w = api.World("FinalStatic", seed=1, aspect=W / H)

    def render_sum(first, count):
        if count == 0:
            return torch.zeros(H, W, 3, dtype=torch.float64)
        p = api.make_params(W, H, count, seed=w.render_seed, spp_offset=first)
        img, _, _ = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, n_threads=2)
        return torch.from_numpy(img * count)

    first, count = partition_spp(SPP, world, rank)
    acc = render_and_reduce(render_sum, first, count, dist)
    dist.barrier()
    if rank == 0:
        whole = render_sum(0, SPP)
        q.put((float((acc - whole).abs().max()), float(whole.abs().max()), first, count))
    dist.destroy_process_group()


def test_spp_partition_reduce_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err, mx, first, count = q.get(timeout=10)
    assert (first, count) == (0, 5)
    assert mx > 1.0 and err < 1e-9 * mx + 1e-12
