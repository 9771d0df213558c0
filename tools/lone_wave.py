"""Debug: the cost of a pass for a wave that is alone on the GPU (the end-of-launch drain is made of such passes).
An 8x8 image at 2 spp is one work unit of 128 items = one wave, one fill; with max_depth = d every path that survives
runs d passes, so kernel time against d gives the time per pass of a lone wave."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracing_amd import api
ctx = api.Context(0)
for scene in sys.argv[1:] or ["FinalStatic"]:
    w = api.World(scene, seed=1, aspect=1.0)
    ctx.upload(w.scene)
    acc = torch.zeros(8, 8, 3, dtype=torch.float32, device="cuda")
    for d in (1, 2, 4, 8, 16, 32, 64, 100, 200, 400):
        p = api.make_params(8, 8, 2, seed=w.render_seed, max_depth=d)
        ctx.render_accum_device(w.camera, p, acc.data_ptr())
        ts = sorted(ctx.render_accum_device(w.camera, p, acc.data_ptr()).kernel_ms for _ in range(5))
        st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
        print(scene, "max_depth", d, "kernel_us %.1f" % (ts[2] * 1e3), "rays", int(st.rays_traced), flush=True)
