"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/liboracle.so).

The reference (Rust) cannot be built or run in this image and its own tests hold no vector for
this path beyond five Vec3 known answers, so these fixtures are AUTHORED by the
build: inputs + the f64 oracle's outputs. They pin (a) the oracle against accidental change and
(b) the HIP path against the oracle without needing the oracle at test time.

Run from the repo root:  python tests/golden/make_golden.py
Scenes come from the host scene factory (librtx_hip.so, rth_world_create, no GPU needed); each
fixture stores a checksum of the flattened scene so a test can tell a scene-factory change from a
kernel regression.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as orc  # noqa: E402
from raytracing_amd import api  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NOISE_SCENES = ["ValueNoise1", "ValueNoise2", "ValueNoisePink", "ValueTurbulence", "ValueMarbled",
                "PerlinNoise1", "PerlinNoise2", "PerlinNoisePink", "PerlinTurbulence", "PerlinMarbled"]


def scene_checksum(world) -> str:
    s = world.scene
    h = hashlib.sha256()
    a = world.arrays()
    for k in ("c0", "dc", "radius", "material_id"):
        h.update(np.ascontiguousarray(a[k]).tobytes())
    h.update(bytes(memoryview((api.Material * s.n_materials).from_address(
        api.C.addressof(s.materials.contents))))) if s.n_materials else None
    if s.n_textures:
        h.update(bytes(memoryview((api.Texture * s.n_textures).from_address(api.C.addressof(s.textures.contents)))))
    if s.n_noise_tables:
        h.update(bytes(memoryview((api.NoiseTable * s.n_noise_tables).from_address(
            api.C.addressof(s.noise_tables.contents)))))
    h.update(bytes(memoryview(world.camera)))
    return h.hexdigest()


def make_rays(origin, direction, time=None):
    r = np.zeros(len(origin), api.ray_dtype)
    r["origin"] = origin
    r["dir"] = direction
    if time is not None:
        r["time"] = time
    return r


def hits_to_f32(h):
    o = np.zeros(len(h), api.hit_dtype)
    for k in ("hit", "sphere_id", "material_id", "front_face"):
        o[k] = h[k]
    o["t"] = h["t"]
    o["point"] = h["point"]
    o["normal"] = h["normal"]
    return o


def ray_cases(world, rng, n_primary, with_time):
    """primary, secondary, inside-glass, grazing and from-inside-overlap rays for one scene."""
    cam = world.camera
    ij = rng.random((n_primary, 2), dtype=np.float32)
    u = rng.random((n_primary, 4), dtype=np.float32)
    prim64 = orc.camera_rays(cam, ij, u)
    prim = make_rays(prim64["origin"], prim64["dir"], prim64["time"] if with_time else None)
    h1 = orc.intersect_batch(world.scene, prim)
    u2 = rng.random((n_primary, 4), dtype=np.float32)
    sc = orc.scatter_batch(world.scene, prim, hits_to_f32(h1), u2)
    m = h1["hit"] == 1
    sec = make_rays(sc["origin"][m], sc["dir"][m], sc["time"][m])
    a = world.arrays()
    n_sph = len(a["radius"])
    # grazing: aim at the limb of random small spheres, offset by r*(1 +- delta)
    k = 1024
    ids = rng.integers(1, n_sph, k)
    o = np.tile(np.array(cam.origin[:], np.float64), (k, 1))
    c = a["c0"][ids].astype(np.float64)
    to_c = c - o
    dist = np.linalg.norm(to_c, axis=1, keepdims=True)
    fwd = to_c / dist
    perp = np.cross(fwd, rng.normal(size=(k, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    delta = rng.choice([-3e-2, -1e-2, -4e-3, 4e-3, 1e-2, 3e-2], k)[:, None]
    target = c + perp * a["radius"][ids][:, None] * (1 + delta)
    graz = make_rays(o, (target - o) * rng.uniform(0.3, 2.0, (k, 1)))
    # inside spheres: origins strictly inside random spheres (far root, back face)
    ids = rng.integers(1, n_sph, k)
    d = rng.normal(size=(k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    inside_o = a["c0"][ids] + d * (a["radius"][ids][:, None] * rng.uniform(0, 0.95, (k, 1)))
    d2 = rng.normal(size=(k, 3)) * rng.uniform(0.2, 3.0, (k, 1))
    ins = make_rays(inside_o, d2)
    return {"primary": prim, "secondary": sec, "grazing": graz, "inside": ins}, (prim, h1, u2)


def main():
    rng = np.random.default_rng(20240610)
    # ---------------- ray + scatter batches
    for name, ntime in (("FinalStatic", False), ("Random", True), ("Balls", False)):
        w = api.World(name, seed=1, aspect=1200 / 800)
        cases, (prim, h1, u2) = ray_cases(w, rng, 4096, ntime)
        out = {"checksum": scene_checksum(w), "t_min": 1e-3}
        for cname, rays in cases.items():
            h = orc.intersect_batch(w.scene, rays, 1e-3, bvh_first=w.bvh_first, bvh_seed=7)
            h_flat = orc.intersect_batch(w.scene, rays, 1e-3)
            # the reference's BVH and the flat list must agree (same closest hit) on these rays
            assert np.array_equal(h["sphere_id"], h_flat["sphere_id"]) and np.array_equal(h["t"], h_flat["t"])
            out[f"{cname}_rays"] = rays
            out[f"{cname}_hits"] = h
        # scatter: primary hits + forced branches
        hits32 = hits_to_f32(h1)
        sc = orc.scatter_batch(w.scene, prim, hits32, u2)
        out["scatter_rays"] = prim
        out["scatter_hits"] = hits32
        out["scatter_u"] = u2
        out["scatter_out"] = sc
        # forced: rays from inside spheres (TIR for glass, back-face for others), grazing incidence
        ins = cases["inside"]
        h_in = hits_to_f32(orc.intersect_batch(w.scene, ins))
        u3 = rng.random((len(ins), 4), dtype=np.float32)
        out["scatter_in_rays"] = ins
        out["scatter_in_hits"] = h_in
        out["scatter_in_u"] = u3
        out["scatter_in_out"] = orc.scatter_batch(w.scene, ins, h_in, u3)
        gz = cases["grazing"]
        h_gz = hits_to_f32(orc.intersect_batch(w.scene, gz))
        u4 = rng.random((len(gz), 4), dtype=np.float32)
        out["scatter_gz_rays"] = gz
        out["scatter_gz_hits"] = h_gz
        out["scatter_gz_u"] = u4
        out["scatter_gz_out"] = orc.scatter_batch(w.scene, gz, h_gz, u4)
        np.savez_compressed(os.path.join(OUT, f"rays_{name}.npz"), **out)
        print(name, {k: len(v) for k, v in cases.items()})
    # ---------------- camera
    w = api.World("Random", seed=1, aspect=1200 / 800)
    ij = rng.random((1024, 2), dtype=np.float32)
    u = rng.random((1024, 4), dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "camera_Random.npz"), checksum=scene_checksum(w), ij=ij, u=u,
                        rays=orc.camera_rays(w.camera, ij, u))
    # ---------------- textures: every noise scene + the checker ground
    tex = {}
    pts = np.concatenate([rng.uniform(-12, 12, (768, 3)), rng.uniform(-1000, 1000, (128, 3)),
                          rng.uniform(-0.5, 0.5, (128, 3))]).astype(np.float32)
    tex["points"] = pts
    for name in NOISE_SCENES + ["Random", "BirdsEyeView"]:
        w = api.World(name, seed=1)
        s = w.scene
        tid = s.materials[s.material_id[0]].texture_id  # the ground's texture
        tex[f"{name}_checksum"] = scene_checksum(w)
        tex[f"{name}_tex_id"] = tid
        tex[f"{name}_values"] = orc.texture_batch(s, tid, pts)
    np.savez_compressed(os.path.join(OUT, "textures.npz"), **tex)
    # ---------------- images
    imgs = {}
    for name, W, H, spp in (("FinalStatic", 120, 68, 2048), ("Random", 120, 68, 2048), ("PerlinMarbled", 120, 68, 1024),
                            ("PerlinTurbulence", 96, 54, 512), ("TwoSpheres", 96, 54, 512)):
        w = api.World(name, seed=1, aspect=W / H)
        p = api.make_params(W, H, spp, seed=w.render_seed)
        img, sec, rays = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, bvh_first=w.bvh_first, bvh_seed=7,
                                    n_threads=8)
        imgs[f"{name}_checksum"] = scene_checksum(w)
        imgs[f"{name}_shape"] = np.array([W, H, spp])
        imgs[f"{name}_seed"] = np.uint64(w.render_seed)
        imgs[f"{name}_img"] = img.astype(np.float32)
        imgs[f"{name}_rays"] = np.uint64(rays)
        print(name, W, H, spp, f"{sec:.1f}s", rays / (W * H * spp))
    # short path-level parity image (few spp: compares sample for sample)
    w = api.World("FinalStatic", seed=1, aspect=100 / 56)
    p = api.make_params(100, 56, 16, seed=w.render_seed)
    img, _, rays = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, bvh_first=w.bvh_first, n_threads=8)
    imgs["FinalStatic16_img"] = img.astype(np.float32)
    imgs["FinalStatic16_rays"] = np.uint64(rays)
    imgs["FinalStatic16_checksum"] = scene_checksum(w)
    np.savez_compressed(os.path.join(OUT, "images.npz"), **imgs)
    # ---------------- RNG
    c = rng.integers(0, 2**32, (256, 4), dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "philox.npz"), seed=np.uint64(0x0123456789ABCDEF), counters=c,
                        out=orc.philox(0x0123456789ABCDEF, c))
    print("done")


if __name__ == "__main__":
    main()
