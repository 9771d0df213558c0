"""Debug: random caller-built scenes (three materials, checker / solid textures, motion, big spheres) rendered small
and compared with the oracle sample for sample (MODE_COUNTER): image RMSE after gamma and ray count."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import api
from oracle import binding as orc


def rmse_gamma(a, b):  # as tests/conftest.py
    ga, gb = np.sqrt(np.clip(a, 0, 1).astype(np.float64)), np.sqrt(np.clip(b, 0, 1).astype(np.float64))
    return float(np.sqrt(((ga - gb) ** 2).mean()))


ctx = api.Context(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([2, 9, 40, 150, 600]))
    c0 = np.empty((n, 3), np.float32)
    c0[:, 0] = rng.uniform(-8, 8, n); c0[:, 1] = rng.uniform(0.2, 3, n); c0[:, 2] = rng.uniform(-8, 8, n)
    rng_p = np.random.default_rng(10**6 + seed0 + case)  # (its own stream: the scenes of earlier runs keep their seeds)
    if rng_p.random() < 0.5:  # some or all centres in one plane (the planar scan list when enough of them stay static)
        c0[rng_p.random(n) < rng_p.uniform(0.3, 1.0), int(rng_p.integers(0, 3))] = np.float32(rng_p.uniform(0.2, 3))
    rad = rng.uniform(0.1, 0.6, n).astype(np.float32)
    dc = np.zeros((n, 3), np.float32)
    mode = int(rng.integers(0, 4))
    if mode:
        ax = rng.permutation(3)[:mode]
        mv = rng.random(n) < 0.4
        dc[np.ix_(mv, ax)] = rng.uniform(-0.4, 0.4, (int(mv.sum()), mode))
    c0[0] = (0, -1000, 0); rad[0] = 1000; dc[0] = 0
    checker = api.solid_texture((0.9, 0.9, 0.9)); checker.kind = api.RT_TEX_CHECKER
    checker.color_b = (api.C.c_float * 3)(0.2, 0.3, 0.1); checker.freq = 10.0
    texs = [api.solid_texture((0.5, 0.5, 0.5)), checker, api.solid_texture((0.7, 0.2, 0.2))]
    mats = [(api.RT_MAT_LAMBERTIAN, (0, 0, 0), 0, 0, 1), (api.RT_MAT_LAMBERTIAN, (0, 0, 0), 0, 0, 0), (api.RT_MAT_LAMBERTIAN, (0, 0, 0), 0, 0, 2),
            (api.RT_MAT_METAL, (0.8, 0.8, 0.6), float(rng.uniform(0, 0.5)), 0, -1), (api.RT_MAT_DIELECTRIC, (1, 1, 1), 0, 1.5, -1)]
    mid = rng.integers(1, 5, n).astype(np.uint32)
    mid[0] = 0
    fs = api.FlatScene(c0, rad, mid, mats, texs, dc=dc)
    ctx.upload(fs.scene)
    cam = api.camera_build((13, 2 + float(rng.uniform(0, 4)), 3), (0, 0.5, 0), vfov_degrees=float(rng.uniform(20, 70)), aspect_ratio=72 / 48,
                           aperture=float(rng.choice([0.0, 0.1])), focus_dist=10.0, shutter=(0.0, 1.0) if mode else None)
    p = api.make_params(72, 48, 8, max_depth=int(rng.choice([3, 10, 50])), seed=int(rng.integers(0, 2 ** 31)))
    img, st = ctx.render(cam, p)
    ref, _, rays_ref = orc.render(fs.scene, cam, p, mode=orc.MODE_COUNTER, n_threads=8)
    r = rmse_gamma(img, ref)
    med = float(np.median(np.abs(img - ref)))
    ok = r < 0.03 and med < 1e-5 and abs(int(st.rays_traced) - rays_ref) <= 0.01 * rays_ref
    bad += not ok
    print("case" if ok else "CASE FAILED", seed0 + case, "n", n, "motion axes", mode, "depth", p.max_depth, "rmse %.5f median %.2e" % (r, med),
          "rays", int(st.rays_traced), rays_ref, flush=True)
print("failed:", bad)
