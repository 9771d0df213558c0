"""Generates tests/golden/near_root_just_ahead.npz: the scene and rays of tools/fuzz_parity.py case 31702 (3000 random
spheres, ext 10, 1920 of them in the planar list), kept because ray 4686 of it broke a round-3 build: its origin lies
inside one sphere and 4e-5 radii OUTSIDE another whose near root is 7e-5 above t_min.  The fp32 scan computes that near
root as slightly behind the scan's start (the scan-side radius is widened), i.e. the sphere looks like one the ray starts
inside; a build that keyed such a sphere by its FAR root (0.43) let a farther sphere (0.0038) win.  The key of an
"origin inside" candidate must stay +0.  Rays: the failing one plus the 255 before it.  Expected hits come from the oracle
at test time (tests/test_gpu_parity.py::test_near_root_just_ahead_of_the_scan_start)."""
import os
import numpy as np
case = 31702
rng = np.random.default_rng(case)
n = int(rng.choice([1, 3, 15, 17, 63, 200, 700, 3000]))
ext = float(rng.choice([1.0, 10.0, 50.0]))
c0 = rng.uniform(-ext, ext, (n, 3)).astype(np.float32)
rad = (ext * 10 ** rng.uniform(-2.5, -0.7, n)).astype(np.float32)
rng_p = np.random.default_rng(10**6 + case)
assert rng_p.random() < 0.5
planar = int(rng_p.integers(0, 3))
c0[rng_p.random(n) < rng_p.uniform(0.3, 1.0), planar] = np.float32(rng_p.uniform(-ext, ext))
mode = int(rng.integers(0, 4)); assert mode == 0
n_big = int(rng.integers(0, 3)); assert n_big == 0
m = 16384
o = rng.uniform(-2 * ext, 2 * ext, (m, 3))
tgt = c0[rng.integers(0, n, m)] + rng.normal(size=(m, 3)) * rad[rng.integers(0, n, m), None]
d = tgt - o
d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.3, 3.0, (m, 1))
t = rng.random(m)
sel = slice(4686 - 255, 4686 + 1)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "near_root_just_ahead.npz")
np.savez_compressed(out, c0=c0, radius=rad, origin=o[sel].astype(np.float32), dir=d[sel].astype(np.float32), time=t[sel].astype(np.float32),
                    t_min=np.float32(1e-3 * ext / 10))
print(out, n, ext, planar)
