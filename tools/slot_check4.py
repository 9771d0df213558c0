import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
W, H = 1200, 800
def R(spp, off, force=0):
    p = api.make_params(W, H, spp, seed=w.render_seed, spp_offset=off)
    p.reserved = force
    img, st = ctx.render(w.camera, p)
    return img.astype(np.float64), st.rays_traced
ones = sum(R(1, i)[0] for i in range(8)) / 8
rays1 = sum(R(1, i)[1] for i in range(8))
for f in (1, 2, 3, 4, 8):
    img, rays = R(8, 0, f)
    d = np.abs(img - ones).max(2)
    print("force", f, "pixels differing", int((d > 1e-4).sum()), "rays", rays, "vs", rays1)
