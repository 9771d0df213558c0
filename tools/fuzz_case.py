"""Debug: one case of tools/fuzz_parity.py (same seeds) with every mismatching ray printed: the ray, the two spheres, their
f64 roots.  tools/fuzz_case.py <case number> [lib.so]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 2:
    os.environ["RTX_HIP_LIB"] = os.path.join(ROOT, "raytracing_amd", sys.argv[2])
import numpy as np
from raytracing_amd import api
from oracle import binding as orc
if len(sys.argv) > 2:
    api.allow_lib_override()
ctx = api.Context(0)
case = int(sys.argv[1])
rng = np.random.default_rng(case)
n = int(rng.choice([1, 3, 15, 17, 63, 200, 700, 3000]))
ext = float(rng.choice([1.0, 10.0, 50.0]))
c0 = rng.uniform(-ext, ext, (n, 3)).astype(np.float32)
rad = (ext * 10 ** rng.uniform(-2.5, -0.7, n)).astype(np.float32)
rng_p = np.random.default_rng(10**6 + case)
if rng_p.random() < 0.5:
    planar = int(rng_p.integers(0, 3))
    c0[rng_p.random(n) < rng_p.uniform(0.3, 1.0), planar] = np.float32(rng_p.uniform(-ext, ext))
dc = np.zeros((n, 3), np.float32)
mode = int(rng.integers(0, 4))
if mode:
    ax = rng.permutation(3)[:mode]
    mv = rng.random(n) < 0.5
    dc[np.ix_(mv, ax)] = rng.uniform(-0.05 * ext, 0.05 * ext, (int(mv.sum()), mode))
n_big = int(rng.integers(0, 3))
for k in range(min(n_big, n)):
    rad[k] = ext * float(rng.uniform(20, 200)); c0[k] = (0, -rad[k] - ext * float(rng.uniform(0, 1)), 0) if k == 0 else c0[k] * 30
mats = [(api.RT_MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0, 0, 0)]
fs = api.FlatScene(c0, rad, np.zeros(n, np.uint32), mats, [api.solid_texture((0.5, 0.5, 0.5))], dc=dc)
ctx.upload(fs.scene)
print("layout", ctx.debug_scan_layout())
m = 16384
rays = np.zeros(m, api.ray_dtype)
o = rng.uniform(-2 * ext, 2 * ext, (m, 3))
tgt = c0[rng.integers(0, n, m)] + rng.normal(size=(m, 3)) * rad[rng.integers(0, n, m), None]
d = tgt - o
rays["origin"] = o
rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.3, 3.0, (m, 1))
rays["time"] = rng.random(m)
t_min = 1e-3 * ext / 10
got = ctx.intersect_batch(rays, t_min)
exp = orc.intersect_batch(fs.scene, rays, t_min)
bad = np.nonzero((got["sphere_id"] != exp["sphere_id"]) | (got["hit"] != exp["hit"]))[0]
print("mismatches", bad)
def roots(i, k):
    o = rays["origin"][i].astype(np.float64); dd = rays["dir"][i].astype(np.float64); tm = float(rays["time"][i])
    c = c0[k].astype(np.float64) + tm * dc[k].astype(np.float64); r = float(rad[k])
    oc = o - c; a = dd @ dd; hb = oc @ dd; cc = oc @ oc - r * r
    disc = hb * hb - a * cc
    if disc < 0: return None
    s = np.sqrt(disc)
    return ((-hb - s) / a, (-hb + s) / a, "|oc|/r %.6f" % (np.linalg.norm(oc) / r), "len d %.4f" % np.sqrt(a))
for i in bad:
    print("ray", i, rays[i], "t_min", t_min)
    for who, h in (("got", got[i]), ("exp", exp[i])):
        k = int(h["sphere_id"])
        print(" ", who, "hit", h["hit"], "id", k, "t", h["t"], "c", c0[k] if k >= 0 else None, "r", rad[k] if k >= 0 else None, "roots", roots(i, k) if k >= 0 else None)
    # everything the ray's line meets, by near root
    allr = [(roots(i, k), k) for k in range(n)]
    allr = sorted([(r[0], r[1], k, r[2]) for r, k in allr if r is not None and r[1] >= t_min])
    for r in allr[:8]: print("   cand id %d near %.9g far %.9g %s r %.4g" % (r[2], r[0], r[1], r[3], rad[r[2]]))
