"""Debug: random caller-built scenes (sizes, radii over two decades, motion on random axes, big spheres, centres in a common plane) and random ray
batches through rt_intersect_batch against the oracle, with the tests' bar (tests/helpers.py::check_hits)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import api
from oracle import binding as orc
from helpers import check_hits

ctx = api.Context(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
worst = {}
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 3, 15, 17, 63, 200, 700, 3000]))
    ext = float(rng.choice([1.0, 10.0, 50.0]))
    c0 = rng.uniform(-ext, ext, (n, 3)).astype(np.float32)
    rad = (ext * 10 ** rng.uniform(-2.5, -0.7, n)).astype(np.float32)
    rng_p = np.random.default_rng(10**6 + seed0 + case)  # (its own stream: the scenes of earlier runs keep their seeds)
    planar = -1
    if rng_p.random() < 0.5:  # some or all centres in one plane: the planar scan list when >= 160 of them stay static
        planar = int(rng_p.integers(0, 3))
        c0[rng_p.random(n) < rng_p.uniform(0.3, 1.0), planar] = np.float32(rng_p.uniform(-ext, ext))
    dc = np.zeros((n, 3), np.float32)
    mode = int(rng.integers(0, 4))
    if mode:  # 1: one axis, 2: two axes, 3: all
        ax = rng.permutation(3)[:mode]
        mv = rng.random(n) < 0.5
        dc[np.ix_(mv, ax)] = rng.uniform(-0.05 * ext, 0.05 * ext, (int(mv.sum()), mode))
    n_big = int(rng.integers(0, 3))
    for k in range(min(n_big, n)):
        rad[k] = ext * float(rng.uniform(20, 200)); c0[k] = (0, -rad[k] - ext * float(rng.uniform(0, 1)), 0) if k == 0 else c0[k] * 30
    mats = [(api.RT_MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0, 0, 0)]
    fs = api.FlatScene(c0, rad, np.zeros(n, np.uint32), mats, [api.solid_texture((0.5, 0.5, 0.5))], dc=dc)
    ctx.upload(fs.scene)
    planar_slots = ctx.debug_scan_layout()["planar_slots"]
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
    try:
        try:
            st = check_hits(got, exp, 1.0, 1e-5, max_excluded=0.001)
        except AssertionError as e:
            if "too many rays excluded" not in str(e):
                raise
            st = check_hits(got, exp, 1.0, 1e-5, max_excluded=0.5)  # not a mismatch: check the rest of the batch anyway
            st = dict(st, note="more than 0.1 % of the rays within the margin of a decision")
    except AssertionError as e:
        print("CASE", seed0 + case, "n", n, "ext", ext, "motion axes", mode, "big", n_big, "plane axis", planar, "FAILED:", str(e)[:300], flush=True)
        continue
    print("case", seed0 + case, "n", n, "ext", ext, "motion axes", mode, "big", n_big, "plane axis", planar, "planar slots", planar_slots, st, "hits", int(exp["hit"].sum()), flush=True)
