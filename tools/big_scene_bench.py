"""Kernel time vs sphere count on caller-built scenes (n small spheres in a box + ground): how far from linear in n
the hierarchical culling is. Development aid. Usage: RTX_HIP_LIB=... python tools/big_scene_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api
api.allow_lib_override()  # A/B tooling: RTX_HIP_LIB names the build under test
ctx = api.Context(0)
for n in (512, 2048, 8192, 16384):
    rng = np.random.default_rng(n)
    side = (n / 512) ** 0.5 * 12.0  # constant density on the ground plane
    c0 = np.empty((n, 3), np.float32)
    c0[:, 0] = rng.uniform(-side, side, n); c0[:, 1] = 0.2; c0[:, 2] = rng.uniform(-side, side, n)
    rad = np.full(n, 0.2, np.float32)
    c0[0] = (0, -1000, 0); rad[0] = 1000
    mats = [(api.RT_MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0, 0, 0), (api.RT_MAT_METAL, (0.8, 0.8, 0.6), 0.1, 0, -1),
            (api.RT_MAT_DIELECTRIC, (1, 1, 1), 0, 1.5, -1)]
    mid = rng.integers(0, 3, n).astype(np.uint32); mid[0] = 0
    fs = api.FlatScene(c0, rad, mid, mats, [api.solid_texture((0.5, 0.5, 0.5))])
    ctx.upload(fs.scene)
    cam = api.camera_build((13, 2, 3), (0, 0, 0), vfov_degrees=20.0, aspect_ratio=1.5, aperture=0.1, focus_dist=10.0)
    p = api.make_params(600, 400, 64, seed=1)
    ctx.render(cam, p)
    _, st = ctx.render(cam, p)
    print(f"n={n:6d} kernel_ms {st.kernel_ms:8.2f} rays/sample {st.rays_traced/st.samples:.2f} ns per ray {st.kernel_ms*1e6/st.rays_traced:.3f}", flush=True)
