import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api
from oracle import binding as orc
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
W, H = 1200, 800
p8 = api.make_params(W, H, 8, seed=w.render_seed)
a, _ = ctx.render(w.camera, p8)
parts = [ctx.render(w.camera, api.make_params(W, H, 1, seed=w.render_seed, spp_offset=i))[0].astype(np.float64) for i in range(8)]
ones = sum(parts) / 8
ref, _, _ = orc.render(w.scene, w.camera, p8, mode=0, bvh_first=w.bvh_first, n_threads=16)
def cnt(x, y, tol=1e-4): return int((np.abs(x - y).max(2) > tol).sum())
print("8spp-call vs 8x1spp:", cnt(a, ones), " 8spp-call vs oracle:", cnt(a, ref), " 8x1spp vs oracle:", cnt(ones, ref))
# which sample indices differ between the 8-spp call and singles? render pairs (2 spp per call => chunks of 1 unless big)
for i in range(0, 8, 2):
    two, _ = ctx.render(w.camera, api.make_params(W, H, 2, seed=w.render_seed, spp_offset=i))
    print("samples", i, i + 1, "2spp-call vs singles:", cnt(two, (parts[i] + parts[i + 1]) / 2))
