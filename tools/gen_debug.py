"""Debug: replay tests/test_gpu_parity.py::test_scattered_ray_generations for one scene and print every mismatching ray
with the f64 roots of the spheres involved."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import api
from oracle import binding as orc

name = sys.argv[1] if len(sys.argv) > 1 else "FinalStatic"
ctx = api.Context(0)
w = api.World(name, seed=1, aspect=1.5)
A = w.arrays()
rng = np.random.default_rng(3)
n = 1 << 17
ctx.upload(w.scene)
rays = ctx.camera_rays(w.camera, rng.random((n, 2), dtype=np.float32), rng.random((n, 4), dtype=np.float32))
def roots(ray, sid):
    o = ray["origin"].astype(np.float64); d = ray["dir"].astype(np.float64)
    c = A["c0"][sid].astype(np.float64) + float(ray["time"]) * A["dc"][sid].astype(np.float64); r = float(A["radius"][sid])
    oc = o - c; a = d @ d; hb = oc @ d; cc = oc @ oc - r * r; disc = hb * hb - a * cc
    return dict(sid=int(sid), disc=disc, disc_rel=disc / a / max(oc @ oc, r * r), t=((-hb - np.sqrt(max(disc, 0))) / a, (-hb + np.sqrt(max(disc, 0))) / a), oc=np.sqrt(oc @ oc), r=r, len=np.sqrt(a))
for gen in range(5):
    exp = orc.intersect_batch(w.scene, rays)
    got = ctx.intersect_batch(rays)
    bad = np.nonzero((got["hit"] != exp["hit"]) | (got["sphere_id"] != exp["sphere_id"]) | (got["front_face"] != exp["front_face"]))[0]
    print("gen", gen, "n", len(rays), "mismatches", len(bad))
    for i in bad[:6]:
        print("  ray", i, rays[i], "\n   got", got[i], "\n   exp", exp[i])
        for sid in {int(got["sphere_id"][i]), int(exp["sphere_id"][i])}:
            if sid >= 0: print("   ", roots(rays[i], sid))
    m = got["hit"] == 1
    u = rng.random((int(m.sum()), 4), dtype=np.float32)
    nrm = got["normal"][m].astype(np.float64)
    tgt = -nrm + rng.normal(size=nrm.shape) * 10.0 ** rng.uniform(-3.5, -1.0, (len(nrm), 1))
    tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    aim = rng.random(len(nrm)) < 0.5
    u[aim, 0] = np.clip((1.0 - tgt[aim, 2]) / 2.0, 0.0, np.nextafter(np.float32(1), np.float32(0)))
    u[aim, 1] = np.mod(np.arctan2(tgt[aim, 1], tgt[aim, 0]) / (2 * np.pi), 1.0).astype(np.float32) % np.float32(1)
    sc = ctx.scatter_batch(rays[m], got[m], u)
    rays = np.zeros(int(m.sum()), api.ray_dtype)
    rays["origin"], rays["dir"], rays["time"] = sc["origin"], sc["dir"], sc["time"]
