"""Generates tests/golden/reference_images.npz: north_star check (2) made DIRECT.

For FinalStatic, Random and PerlinTurbulence at BASELINE configs[0] geometry (400x225) the CPU oracle renders the
image the way the REFERENCE samples it (oracle MODE_REFERENCE: f64, BVH, per-row sequential Pcg64Mcg seeded
`(seed+1)*y` as main.rs:71, rand_distr-style rejection samplers for the unit sphere / disc as vec3.rs:53-62, the
Dielectric uniform drawn only when not TIR as material.rs:101-102) -- NOT with the kernel's counter RNG.  Two
INDEPENDENT renders (render seeds SEED_A / SEED_B) are stored per (scene, spp): their RMSE after gamma is the
Monte-Carlo noise floor from which tests/test_gpu_reference_images.py derives its thresholds, so no tolerance
there is hand-picked.

Storage: gamma-encoded (sqrt, what `Screen::encode` applies, screen.rs:36-40) and quantised to uint16 (step 1.5e-5:
three orders of magnitude below the noise floor), to keep the fixture at a few MB.

Run from the repo root (about 10 minutes on 8 cores):  python tests/golden/make_reference_images.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as orc  # noqa: E402
from raytracing_amd import api  # noqa: E402
from helpers import scene_checksum  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
W, H = 400, 225
SCENES = ("FinalStatic", "Random", "PerlinTurbulence")
SPPS = (1000, 4096)
SEED_A, SEED_B = 0x5EED0A, 0x5EED0B


def quantise(img):
    return np.round(np.sqrt(np.clip(img, 0.0, 1.0)) * 65535.0).astype(np.uint16)


def main():
    threads = len(os.sched_getaffinity(0))
    out = {"width": W, "height": H, "seed_a": np.uint64(SEED_A), "seed_b": np.uint64(SEED_B)}
    for name in SCENES:
        w = api.World(name, seed=1, aspect=W / H)
        out[f"{name}_checksum"] = scene_checksum(api, w)
        for spp in SPPS:
            for tag, seed in (("a", SEED_A), ("b", SEED_B)):
                p = api.make_params(W, H, spp, seed=seed)
                t0 = time.time()
                img, sec, rays = orc.render(w.scene, w.camera, p, mode=orc.MODE_REFERENCE, bvh_first=w.bvh_first,
                                            bvh_seed=7, n_threads=threads, precision=64)
                assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= 1
                out[f"{name}_{spp}_{tag}"] = quantise(img)
                out[f"{name}_{spp}_{tag}_mean"] = img.mean((0, 1))  # per-channel mean linear radiance, f64
                out[f"{name}_{spp}_{tag}_rays"] = np.uint64(rays)
                print(name, spp, tag, f"{time.time() - t0:.1f}s rays/sample {rays / (W * H * spp):.4f} "
                      f"mean {img.mean((0, 1))}", flush=True)
            a = out[f"{name}_{spp}_a"].astype(np.float64) / 65535.0
            b = out[f"{name}_{spp}_b"].astype(np.float64) / 65535.0
            print(name, spp, "noise floor rmse(gamma)(A,B) =", float(np.sqrt(((a - b) ** 2).mean())), flush=True)
    np.savez_compressed(os.path.join(OUT, "reference_images.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
