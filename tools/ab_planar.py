"""Debug: the planar scan list against the general one (rt_debug_set_planar_list): images, ray counts, kernel time."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytracing_amd import api
ctx = api.Context(0)
cases = [("FinalStatic", 100, 56, 16, 50), ("Random", 328, 200, 40, 100), ("Balls", 64, 36, 64, 100), ("PerlinNoise2", 96, 64, 16, 20),
         ("FinalStatic", 1200, 800, 100, 100), ("Random", 1200, 800, 50, 100)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for name, W, H, spp, depth in cases:
    w = api.World(name, seed=1, aspect=W / H)
    out = []
    for on in (False, True):
        ctx.debug_set_planar_list(on)
        ctx.upload(w.scene)
        p = api.make_params(W, H, spp, seed=w.render_seed, max_depth=depth)
        img, st = ctx.render(w.camera, p)
        img, st = ctx.render(w.camera, p)
        out.append((img, int(st.rays_traced), st.kernel_ms))
    d = np.abs(out[0][0] - out[1][0])
    print(name, W, H, spp, "depth", depth, "pixels differing", int((d.max(2) > 0).sum()), "max", float(d.max()), "rays", out[0][1], out[1][1],
          "kernel ms general %.2f planar %.2f" % (out[0][2], out[1][2]), flush=True)
ctx.debug_set_planar_list(True)
