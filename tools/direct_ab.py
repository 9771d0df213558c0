"""Debug: the scan path against the direct (all-f64) path on a scene of a few spheres, both against the oracle:
camera rays, then three generations of scattered rays (origins ON surfaces: the rays a render traces)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import api
from oracle import binding as orc
from helpers import check_hits

ctx = api.Context(0)
for scene in sys.argv[1:] or ["PerlinNoise2", "Balls", "TwoSpheres"]:
    w = api.World(scene, seed=1, aspect=1.5)
    rng = np.random.default_rng(3)
    n = 1 << 18
    ctx.upload(w.scene)
    rays = ctx.camera_rays(w.camera, rng.random((n, 2), dtype=np.float32), rng.random((n, 4), dtype=np.float32))
    for gen in range(4):
        exp = orc.intersect_batch(w.scene, rays)
        res = {}
        for direct in (True, False):
            ctx.debug_set_direct(direct)
            ctx.upload(w.scene)
            lay = ctx.debug_scan_layout()
            got = ctx.intersect_batch(rays)
            bad = (got["hit"] != exp["hit"]) | (got["sphere_id"] != exp["sphere_id"]) | (got["front_face"] != exp["front_face"])
            ok = exp["margin_tie"] >= 1
            res[direct] = got
            print(scene, "gen", gen, "direct" if direct else "scan  ", "scan mode", lay["scan"], "mismatches", int(bad.sum()), "of them outside the margin", int((bad & ok).sum()),
                  "excluded", int((~ok).sum()), "hits", int(exp["hit"].sum()))
            if (bad & ok).any():
                i = np.nonzero(bad & ok)[0][:3]
                print("   first:", i, got[i], exp[i][["hit", "sphere_id", "front_face", "t", "margin", "margin_tie"]], rays[i])
        ctx.debug_set_direct(True)
        # next generation: scatter the hits
        got = res[True]
        m = got["hit"] == 1
        sc = ctx.scatter_batch(rays[m], got[m], rng.random((int(m.sum()), 4), dtype=np.float32))
        nxt = np.zeros(int(m.sum()), api.ray_dtype)
        nxt["origin"], nxt["dir"], nxt["time"] = sc["origin"], sc["dir"], sc["time"]
        rays = nxt
        if len(rays) == 0:
            break
