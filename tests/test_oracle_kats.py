"""CPU tests of the oracle: the reference's own known answers + published RNG vectors + golden pins.

The only known-answer arithmetic the reference holds for this path is its five Vec3 doctests
(vec3.rs:27-33, 41-46, 64-70, 87-93, 98-104); they are the oracle's first tests, as SURVEY.md §4 asks.
"""
import numpy as np
import pytest

from conftest import golden, rmse_gamma
from helpers import scene_checksum


# ---- the reference's doctests -------------------------------------------------------------------
def test_doctest_normalized_norm_is_one(orc):  # vec3.rs:27-33
    assert orc.vec3_norm(orc.vec3_normalized([1.0, 2.0, 3.0])) == 1.0


def test_doctest_checked_normalized_zero_is_err(orc):  # vec3.rs:41-46
    rc, _ = orc.vec3_checked_normalized([0.0, 0.0, 0.0])
    assert rc != 0
    rc, v = orc.vec3_checked_normalized([0.0, 3.0, 4.0])
    assert rc == 0 and np.allclose(v, [0.0, 0.6, 0.8])


def test_doctest_map_floor_times_two(orc):  # vec3.rs:64-70
    assert np.array_equal(orc.vec3_map_floor_times2([1.2, 2.3, 3.4]), [2.0, 4.0, 6.0])


def test_doctest_dot(orc):  # vec3.rs:87-93
    assert orc.vec3_dot([4.0, 8.0, 10.0], [9.0, 2.0, 7.0]) == 122.0


def test_doctest_cross(orc):  # vec3.rs:98-104
    assert np.array_equal(orc.vec3_cross([2.0, 3.0, 4.0], [5.0, 6.0, 7.0]), [-3.0, 6.0, -3.0])


# ---- helpers the path uses ------------------------------------------------------------------------
def test_reflect_normalises_first(orc):  # vec3.rs:113-116
    r = orc.vec3_reflect([0.0, -10.0, 0.0], [0.0, 1.0, 0.0])
    assert np.allclose(r, [0.0, 1.0, 0.0])
    r = orc.vec3_reflect([3.0, -3.0, 0.0], [0.0, 1.0, 0.0])
    assert np.allclose(r, np.array([1.0, 1.0, 0.0]) / np.sqrt(2))


def test_refract_snell(orc):  # vec3.rs:118-123
    n = np.array([0.0, 1.0, 0.0])
    for deg in (0.0, 10.0, 35.0, 60.0):
        th = np.radians(deg)
        d = np.array([np.sin(th), -np.cos(th), 0.0])
        r = orc.vec3_refract(d, n, 1 / 1.5)
        assert np.isclose(np.linalg.norm(r), 1.0)
        assert np.isclose(np.sin(th) / 1.5, r[0])  # Snell: sin(theta_t) = eta * sin(theta_i)


def test_lerp_smooth_schlick(orc):  # lib.rs:45-54, material.rs:84-88
    L = orc.load()
    assert L.orc_lerp(0.25, 2.0, 6.0) == 2.0 * 0.75 + 6.0 * 0.25
    for t in (0.0, 0.5, 1.0, 0.3):
        assert np.isclose(L.orc_smooth(t), 6 * t**5 - 15 * t**4 + 10 * t**3)
    assert np.isclose(L.orc_schlick(1.0, 1 / 1.5), ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2)
    assert np.isclose(L.orc_schlick(0.0, 1.5), 1.0)


# ---- counter RNG: Random123 known-answer vectors for philox4x32-10 -----------------------------------
PHILOX_KAT = [
    ((0, 0, 0, 0), 0, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, 0xFFFFFFFFFFFFFFFF, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), 0x299F31D0A4093822,
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,exp", PHILOX_KAT)
def test_philox_random123_kat(orc, ctr, key, exp):
    out = orc.philox(key, np.array([ctr], np.uint32))[0]
    assert tuple(int(x) for x in out) == exp


def test_philox_golden(orc):
    g = golden("philox.npz")
    assert np.array_equal(orc.philox(int(g["seed"]), g["counters"]), g["out"])


# ---- sampling maps have the distributions of rand_distr::UnitSphere / UnitDisc ----------------------
def test_unit_sphere_map_distribution(orc):
    rng = np.random.default_rng(3)
    u = rng.random((20000, 2))
    v = np.array([orc.map_unit_sphere(a, b) for a, b in u[:2000]])
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-12)
    # uniform on S^2: each coordinate is uniform on [-1,1] (Archimedes) -> mean 0, variance 1/3
    assert np.all(np.abs(v.mean(0)) < 0.05) and np.allclose(v.var(0), 1 / 3, atol=0.03)
    ref = orc.pcg_unit_sphere(99, 20000)  # the rejection sampler the reference calls
    assert np.allclose(np.linalg.norm(ref, axis=1), 1.0, atol=1e-12)
    assert np.all(np.abs(ref.mean(0)) < 0.03) and np.allclose(ref.var(0), 1 / 3, atol=0.01)


def test_unit_disk_map_distribution(orc):
    rng = np.random.default_rng(4)
    v = np.array([orc.map_unit_disk(a, b) for a, b in rng.random((2000, 2))])
    r2 = (v[:, :2] ** 2).sum(1)
    assert np.all(r2 <= 1.0 + 1e-12) and np.all(v[:, 2] == 0)
    assert abs(r2.mean() - 0.5) < 0.03  # uniform in the disc: E[r^2] = 1/2
    ref = orc.pcg_unit_disk(5, 20000)
    r2 = (ref[:, :2] ** 2).sum(1)
    assert np.all(r2 <= 1.0) and abs(r2.mean() - 0.5) < 0.01


def test_pcg64mcg_is_deterministic_and_full_width(orc):
    a, b = orc.pcg64mcg(1, 64), orc.pcg64mcg(1, 64)
    assert np.array_equal(a, b) and len(set(a.tolist())) == 64
    assert not np.array_equal(a, orc.pcg64mcg(2, 64))
    assert (a >> np.uint64(63)).sum() > 10  # top bit is exercised


# ---- closest hit: BVH (hit.rs:191-288) == flat list (hit.rs:100-110); golden pins ------------------
@pytest.mark.parametrize("name", ["FinalStatic", "Random", "Balls"])
def test_intersect_golden_and_bvh_agree(api, orc, name):
    g = golden(f"rays_{name}.npz")
    w = api.World(name, seed=1, aspect=1200 / 800)
    assert scene_checksum(api, w) == str(g["checksum"]), "host scene factory changed: regenerate tests/golden"
    for case in ("primary", "secondary", "grazing", "inside"):
        rays, exp = g[f"{case}_rays"], g[f"{case}_hits"]
        flat = orc.intersect_batch(w.scene, rays, 1e-3)
        for k in ("hit", "sphere_id", "material_id", "front_face", "t", "point", "normal"):
            assert np.array_equal(flat[k], exp[k]), (case, k)
        bvh = orc.intersect_batch(w.scene, rays, 1e-3, bvh_first=max(w.bvh_first, 0), bvh_seed=123)
        assert np.array_equal(bvh["sphere_id"], exp["sphere_id"]) and np.array_equal(bvh["t"], exp["t"])
    # the cases do what they are meant to
    assert (g["inside_hits"]["front_face"] == 0).sum() > 0.6 * len(g["inside_hits"])
    # aimed within r*(1 +- 3e-2) of a limb: far tighter margins than generic primary rays
    assert np.median(g["grazing_hits"]["margin"]) < 0.2 * np.median(g["primary_hits"]["margin"])


def test_intersect_edge_cases(api, orc):
    w = api.World("Balls", seed=1)
    # empty batch, ray pointing away, zero-length-free: a ray starting exactly on a surface going out
    assert len(orc.intersect_batch(w.scene, np.zeros(0, api.ray_dtype))) == 0
    r = np.zeros(3, api.ray_dtype)
    r["origin"] = [[0, 0, 1], [0, 0, -0.5], [0, 0, -1]]
    r["dir"] = [[0, 0, 1], [0, 0, 1], [0, 0, 7]]
    h = orc.intersect_batch(w.scene, r)
    assert h["hit"][0] == 0  # looks away from everything
    assert h["hit"][1] == 0  # starts on the glass ball's surface (t = 0 < t_min), leaves it
    assert h["hit"][2] == 1 and h["sphere_id"][2] == 1 and h["front_face"][2] == 0  # centre of the ball: far root
    assert np.isclose(h["t"][2], 0.5 / 7)  # unnormalised direction: t in units of |d|


def test_scatter_golden(api, orc):
    for name in ("FinalStatic", "Random", "Balls"):
        g = golden(f"rays_{name}.npz")
        w = api.World(name, seed=1, aspect=1200 / 800)
        for pre in ("scatter", "scatter_in", "scatter_gz"):
            out = orc.scatter_batch(w.scene, g[f"{pre}_rays"], g[f"{pre}_hits"], g[f"{pre}_u"])
            exp = g[f"{pre}_out"]
            for k in ("scattered", "albedo", "origin", "dir", "time"):
                assert np.array_equal(out[k], exp[k]), (name, pre, k)


def test_scatter_branches_are_covered(api):
    """The forced cases reach TIR, refraction, and Metal's below-surface correction (material.rs:66-70)."""
    g = golden("rays_Balls.npz")  # 1 of its 5 small spheres is glass (config.rs:220)
    w = api.World("Balls", seed=1, aspect=1200 / 800)
    s = w.scene
    mats = np.array([s.materials[i].type for i in range(s.n_materials)])
    hits, out, rays = g["scatter_in_hits"], g["scatter_in_out"], g["scatter_in_rays"]
    m = hits["hit"] == 1
    glass = m & (mats[hits["material_id"]] == api.RT_MAT_DIELECTRIC)
    assert glass.sum() > 20
    # inside glass: refraction changes the direction's tangential part; reflection keeps |d.n|
    d = rays["dir"][glass] / np.linalg.norm(rays["dir"][glass], axis=1, keepdims=True)
    n = hits["normal"][glass]
    cos_i = -(d * n).sum(1)
    sin_i = np.sqrt(np.maximum(0, 1 - cos_i**2))
    tir = 1.5 * sin_i > 1.0
    assert tir.sum() > 5 and (~tir).sum() > 5
    o = out["dir"][glass]
    refl = d - 2 * (d * n).sum(1, keepdims=True) * n
    assert np.allclose(o[tir], refl[tir], atol=1e-6)  # TIR must reflect
    # metal below-surface branch: grazing rays on fuzzy metal
    g = golden("rays_FinalStatic.npz")
    w2 = api.World("FinalStatic", seed=1, aspect=1200 / 800)  # keep alive: `scene` points into it
    s = w2.scene
    mats = np.array([s.materials[i].type for i in range(s.n_materials)])
    hz, oz, rz, uz = g["scatter_gz_hits"], g["scatter_gz_out"], g["scatter_gz_rays"], g["scatter_gz_u"]
    metal = (hz["hit"] == 1) & (mats[hz["material_id"]] == api.RT_MAT_METAL)
    assert metal.sum() > 10
    # recompute the pre-correction direction; where it dips below the surface the output differs from it
    d = rz["dir"][metal] / np.linalg.norm(rz["dir"][metal], axis=1, keepdims=True)
    n = hz["normal"][metal].astype(np.float64)
    refl = d - 2 * (d * n).sum(1, keepdims=True) * n
    from oracle import binding as orc
    fuzz = np.array([s.materials[i].fuzz for i in hz["material_id"][metal]], np.float64)
    S = np.array([orc.map_unit_sphere(float(a), float(b)) for a, b in uz[metal][:, :2]])
    pre = refl + fuzz[:, None] * S
    below = (pre * n).sum(1) <= 0
    assert below.sum() >= 3, "no ray reaches Metal's below-surface correction"
    exp = np.where(below[:, None], pre - 2 * fuzz[:, None] * S, pre)
    assert np.allclose(oz["dir"][metal], exp, atol=1e-6)


def test_textures_golden_and_ranges(api, orc):
    g = golden("textures.npz")
    pts = g["points"]
    for name in ["ValueNoise1", "ValueNoise2", "ValueNoisePink", "ValueTurbulence", "ValueMarbled", "PerlinNoise1",
                 "PerlinNoise2", "PerlinNoisePink", "PerlinTurbulence", "PerlinMarbled", "Random", "BirdsEyeView"]:
        w = api.World(name, seed=1)
        assert scene_checksum(api, w) == str(g[f"{name}_checksum"])
        v = orc.texture_batch(w.scene, int(g[f"{name}_tex_id"]), pts)
        assert np.array_equal(v, g[f"{name}_values"]), name
        assert np.all(v >= 0.0) and np.all(v <= 1.0), name  # Screen::encode would panic otherwise
        if "Noise" in name or "Turb" in name or "Marb" in name:
            assert np.array_equal(v[:, 0], v[:, 1]) and np.array_equal(v[:, 0], v[:, 2])  # white * value
            assert v[:, 0].std() > 0.02


def test_perlin_gradient_table_rows_9_11_15_agree(api, orc):
    """material.rs:441-448: cases 9, 11 and 15 are all -y - z (the code, not its comments)."""
    w = api.World("PerlinNoise1", seed=1)
    tid = w.scene.materials[w.scene.material_id[0]].texture_id
    v = orc.texture_batch(w.scene, tid, np.array([[0.1, 0.2, 0.3], [0.35, 0.2, 0.3]], np.float32))
    assert v.shape == (2, 3) and not np.array_equal(v[0], v[1])


def test_camera_golden(api, orc):
    g = golden("camera_Random.npz")
    w = api.World("Random", seed=1, aspect=1200 / 800)
    assert scene_checksum(api, w) == str(g["checksum"])
    r = orc.camera_rays(w.camera, g["ij"], g["u"])
    for k in ("origin", "dir", "time"):
        assert np.array_equal(r[k], g["rays"][k])
    assert np.all((r["time"] >= 0) & (r["time"] < 1))  # shutter 0..1 (config.rs:118)
    off = np.linalg.norm(r["origin"] - np.array(w.camera.origin[:]), axis=1)
    assert np.all(off <= 0.05 + 1e-7) and off.max() > 0.03  # lens_radius = aperture / 2 = 0.05


def test_render_golden_counter_mode(api, orc):
    g = golden("images.npz")
    w = api.World("FinalStatic", seed=1, aspect=100 / 56)
    assert scene_checksum(api, w) == str(g["FinalStatic16_checksum"])
    p = api.make_params(100, 56, 16, seed=w.render_seed)
    img, _, rays = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, bvh_first=w.bvh_first, n_threads=4)
    assert rays == int(g["FinalStatic16_rays"])
    assert np.array_equal(img.astype(np.float32), g["FinalStatic16_img"])
    # flat list renders the same image as the BVH (closest hit is the same object)
    img2, _, rays2 = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, bvh_first=-1, n_threads=4)
    assert rays2 == rays and np.array_equal(img, img2)
    # spp partition: two halves with spp_offset average to the whole (counter streams are disjoint)
    pa = api.make_params(100, 56, 8, seed=w.render_seed)
    pb = api.make_params(100, 56, 8, seed=w.render_seed, spp_offset=8)
    ia, _, _ = orc.render(w.scene, w.camera, pa, n_threads=4)
    ib, _, _ = orc.render(w.scene, w.camera, pb, n_threads=4)
    assert np.allclose(0.5 * (ia + ib), img, rtol=0, atol=1e-12)


def test_reference_sampler_mode_agrees_statistically(api, orc):
    """Sequential PCG + rejection samplers (the reference's sampling, main.rs:71-87) and the counter RNG +
    direct maps converge to the same image: the RNG replacement is distribution-preserving."""
    w = api.World("FinalStatic", seed=1, aspect=64 / 36)
    p = api.make_params(64, 36, 256, seed=w.render_seed)
    a, _, ra = orc.render(w.scene, w.camera, p, mode=orc.MODE_COUNTER, bvh_first=w.bvh_first, n_threads=8)
    b, _, rb = orc.render(w.scene, w.camera, p, mode=orc.MODE_REFERENCE, bvh_first=w.bvh_first, n_threads=8)
    assert rmse_gamma(a, b) < 0.035  # Monte-Carlo noise of two independent 256-spp renders
    assert abs(ra - rb) / ra < 0.01
    assert abs(a.mean() - b.mean()) < 0.004
    enc, bad = orc.encode(a)
    assert bad == 0 and enc.max() <= 0x00FFFFFF


def test_encode_known_answers(orc):  # screen.rs:26-45
    enc, bad = orc.encode(np.array([[0.0, 0.25, 1.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5]]))
    assert bad == 0
    assert enc[0] == (0 << 16) | (127 << 8) | 255  # 255.99*0.5 = 127.995 -> 127
    assert enc[1] == 0x00FFFFFF
    assert enc[2] == (181 << 16) | (181 << 8) | 181  # 255.99*sqrt(.5) = 181.01
    enc, bad = orc.encode(np.array([[np.nan, 0.5, 0.5], [1.5, 0, 0], [-0.1, 0, 0]]))
    assert bad == 3  # the reference panics on each of these
