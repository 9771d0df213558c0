"""Debug: five renders of three scenes must be bit-identical (image hash and ray count)."""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from raytracing_amd import api
ctx = api.Context(0)
for scene, spp in (("FinalStatic", 200), ("Random", 100), ("PerlinTurbulence", 100)):
    w = api.World(scene, seed=1, aspect=1.5)
    ctx.upload(w.scene)
    hs = set()
    for i in range(5):
        img, st = ctx.render(w.camera, api.make_params(1200, 800, spp, seed=w.render_seed))
        hs.add((hashlib.sha1(img.tobytes()).hexdigest(), int(st.rays_traced)))
    print(scene, "distinct results over 5 renders:", len(hs))
