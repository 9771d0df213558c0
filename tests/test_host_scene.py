"""CPU tests of the host side above the C ABI (raytracing_amd/csrc/host_scene.cpp) and of the ABI itself.

Mirrors the reference's own unit tests (config.rs:407-426: scene-name parsing) and checks the restated
scene factory / camera builder against independent numpy computations of the cited formulas.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- config.rs:407-426 ---------------------------------------------------------------------------
def test_scene_names_right_case(api):  # parse_test::right_case
    assert api.World("Random", seed=1).scene.n_spheres > 400
    assert api.World("TwoSpheres", seed=1).scene.n_spheres == 2


@pytest.mark.parametrize("bad", ["random", "rANDOM", "twospheres", "two-spheres", "two_spheres", "Two_spheres", ""])
def test_scene_names_wrong_case(api, bad):  # parse_test::wrong_case
    with pytest.raises(api.RtError) as e:
        api.World(bad, seed=1)
    assert e.value.code == api.RT_ERR_BAD_ARG and "isn't a Scene" in str(e.value)


def test_scene_variants(api):  # Scene enum, config.rs:85-101 (+ the documented FinalStatic extension)
    names = api.scene_names()
    assert names[:14] == ["Random", "TwoSpheres", "Balls", "BirdsEyeView", "ValueNoise1", "ValueNoise2",
                          "ValueNoisePink", "ValueTurbulence", "ValueMarbled", "PerlinNoise1", "PerlinNoise2",
                          "PerlinNoisePink", "PerlinTurbulence", "PerlinMarbled"]
    assert names[14:] == ["FinalStatic"]
    for n in names:
        w = api.World(n, seed=3)
        assert w.scene.n_spheres >= 1 and w.scene.n_materials >= 1


# ---- config.rs:150-200 ------------------------------------------------------------------------------
def test_random_world_structure(api):
    w = api.World("Random", seed=1)
    a, s = w.arrays(), w.scene
    n = s.n_spheres
    assert 400 < n <= 488 and w.bvh_first == 1
    assert np.array_equal(a["c0"][0], [0, -1000, 0]) and a["radius"][0] == 1000  # ground first
    assert np.array_equal(a["c0"][-3:], [[0, 1, 0], [-4, 1, 0], [4, 1, 0]]) and np.all(a["radius"][-3:] == 1)
    small = slice(1, n - 3)
    assert np.all(a["radius"][small] == np.float32(0.2)) and np.all(a["c0"][small, 1] == np.float32(0.2))
    # no small sphere within 0.9 of (4, 0.2, 0) (config.rs:164)
    assert np.all(np.linalg.norm(a["c0"][small] - [4, 0.2, 0], axis=1) > 0.9)
    mats = np.array([s.materials[i].type for i in a["material_id"]])
    frac = [(mats[small] == t).mean() for t in (0, 1, 2)]
    assert 0.72 < frac[0] < 0.88 and 0.09 < frac[1] < 0.21 and 0.01 < frac[2] < 0.10  # 80 / 15 / 5 %
    moving = np.abs(a["dc"]).sum(1) > 0
    # every small Lambertian moves, only in +y by U[0, 0.5) (config.rs:171); nothing else moves
    assert np.array_equal(moving[small], mats[small] == 0) and not moving[0] and not moving[-3:].any()
    assert np.all(a["dc"][moving][:, [0, 2]] == 0) and np.all((a["dc"][moving][:, 1] >= 0) & (a["dc"][moving][:, 1] < 0.5))
    # ground is the checker of config.rs:152
    t = s.textures[s.materials[a["material_id"][0]].texture_id]
    assert t.kind == api.RT_TEX_CHECKER and t.freq == 10 and np.allclose(t.color_a[:], [0.2, 0.3, 0.1]) and \
        np.allclose(t.color_b[:], [0.9, 0.9, 0.9])
    # metal: albedo in [0.5,1), fuzz in [0,0.5) (config.rs:175-176); glass 1.5
    for i in a["material_id"][small]:
        m = s.materials[i]
        if m.type == 1:
            assert all(0.5 <= c < 1 for c in m.albedo[:]) and 0 <= m.fuzz < 0.5
        elif m.type == 2:
            assert m.ior == 1.5
    c = w.camera
    assert np.allclose(c.origin[:], [13, 2, 3]) and np.isclose(c.lens_radius, 0.05) and c.has_shutter == 1 and \
        (c.shutter_open, c.shutter_close) == (0.0, 1.0)


def test_final_static_shares_layout_with_random(api):
    """The static variant consumes the same RNG draws, so centres/materials match Random for a seed."""
    a, b = api.World("Random", seed=5), api.World("FinalStatic", seed=5)
    A, B = a.arrays(), b.arrays()
    assert np.array_equal(A["c0"], B["c0"]) and np.array_equal(A["radius"], B["radius"])
    assert not np.abs(B["dc"]).any() and b.camera.has_shutter == 0 and np.isclose(b.camera.lens_radius, 0.05)
    assert a.render_seed == b.render_seed
    t = b.scene.textures[b.scene.materials[B["material_id"][0]].texture_id]
    assert t.kind == api.RT_TEX_SOLID and np.allclose(t.color_a[:], [0.5, 0.5, 0.5])
    assert api.World("Random", seed=6).arrays()["c0"].tolist() != A["c0"].tolist()
    assert api.World("Random", seed=None).scene.n_spheres > 400  # entropy seeding (main.rs:21)


def test_small_scenes(api):  # config.rs:201-254
    b = api.World("Balls", seed=1)
    assert b.scene.n_spheres == 6 and np.allclose(b.arrays()["c0"][0], [0, -100.5, -1]) and b.bvh_first == -1
    assert np.allclose(b.camera.origin[:], [-2, 1.5, 1])
    e = api.World("BirdsEyeView", seed=1)
    assert e.scene.n_spheres == 1 and e.arrays()["radius"][0] == 10
    assert e.scene.textures[0].kind == api.RT_TEX_CHECKER and e.scene.textures[0].freq == 2.5


NOISE = {  # config.rs:255-402: (kind, freq, op, lacunarity, gain, layers, marbled(sf, phase, noise_gain))
    "ValueNoise1": (2, 4, 0, 0, 0, 0, None), "ValueNoise2": (2, 4, 0, 0, 0, 0, None),
    "ValueNoisePink": (2, 2, 1, 2, .5, 5, None), "ValueTurbulence": (2, 5, 2, 1.8, .35, 5, None),
    "ValueMarbled": (2, 3, 1, 2, .5, 2, (2, 2.5, .5)),
    "PerlinNoise1": (3, 4, 0, 0, 0, 0, None), "PerlinNoise2": (3, 4, 0, 0, 0, 0, None),
    "PerlinNoisePink": (3, 2, 1, 2, .5, 5, None), "PerlinTurbulence": (3, 5, 2, 2, .5, 7, None),
    "PerlinMarbled": (3, 2, 1, 2, .5, 5, (2, 0, 2.0)),
}


@pytest.mark.parametrize("name", sorted(NOISE))
def test_noise_scenes(api, name):
    kind, freq, op, lac, gain, layers, marb = NOISE[name]
    w = api.World(name, seed=1)
    s = w.scene
    assert s.n_noise_tables == 1 and s.n_spheres == (3 if name.endswith("Noise1") else 2)
    t = s.textures[s.materials[s.material_id[0]].texture_id]
    assert (t.kind, t.freq, t.op) == (kind, freq, op)
    if op:
        assert np.isclose(t.lacunarity, lac) and np.isclose(t.gain, gain) and t.layers == layers
    if marb:
        assert t.marbled == 1 and np.isclose(t.sin_freq, marb[0]) and np.isclose(t.phase, marb[1]) and \
            np.isclose(t.marble_gain, freq * marb[2])  # gain = freq * noise_gain (material.rs:251)
    else:
        assert t.marbled == 0
    tb = s.noise_tables[0]
    perms = np.array(tb.perms[:])
    assert sorted(perms[:256].tolist()) == list(range(256))  # first half: a permutation
    assert perms[256:].max() <= 255
    # second half was copied DURING the shuffle (material.rs:294-299): it is stale, not perms[i & 255]
    assert not np.array_equal(perms[:256], perms[256:])
    rnd = np.array(tb.randoms[:])
    if kind == 2:
        assert np.all((rnd >= 0) & (rnd < 1)) and rnd.std() > 0.2
    else:
        assert not rnd.any()
    assert np.allclose(w.camera.origin[:], [13, 2, 3]) and w.camera.lens_radius == 0 and w.camera.has_shutter == 0


# ---- screen.rs:112-147 ---------------------------------------------------------------------------------
def test_camera_builder_matches_formulas(api):
    o, la = np.array([13.0, 2, 3]), np.array([0.0, 0, 0])
    cam = api.camera_build(o, la, vfov_degrees=20, aspect_ratio=1.5, aperture=0.1, focus_dist=10, shutter=(0, 1))
    theta = np.radians(20) / 2
    hh = 10 * np.tan(theta)
    hw = 1.5 * hh
    w = (o - la) / np.linalg.norm(o - la)
    u = np.cross([0, 1, 0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    assert np.allclose(cam.u[:], u, atol=1e-7) and np.allclose(cam.v[:], v, atol=1e-7)
    assert np.allclose(cam.lower_left[:], o - u * hw - v * hh - 10 * w, atol=1e-6)
    assert np.allclose(cam.horiz[:], 2 * u * hw, atol=1e-6) and np.allclose(cam.vert[:], 2 * v * hh, atol=1e-6)
    assert np.isclose(cam.lens_radius, 0.05) and cam.has_shutter == 1
    # focus_dist defaults to |origin - look_at| (screen.rs:119)
    cam2 = api.camera_build(o, la, vfov_degrees=30, aspect_ratio=1.0)
    hh2 = np.linalg.norm(o - la) * np.tan(np.radians(15))
    assert np.allclose(np.linalg.norm(cam2.vert[:]), 2 * hh2, rtol=1e-6) and cam2.lens_radius == 0


@pytest.mark.parametrize("kw,msg", [
    (dict(origin=[0, 0, 0], look_at=[0, 0, 0]), "origin and look_at"),
    (dict(origin=[0, 1, 0], look_at=[0, 0, 0]), "parallel"),
    (dict(origin=[1, 1, 0], look_at=[0, 0, 0], view_up=[0, 0, 0]), "view_up"),
    (dict(origin=[1, 1, 0], look_at=[0, 0, 0], aperture=-1), "aperture"),
    (dict(origin=[1, 1, 0], look_at=[0, 0, 0], vfov_degrees=0), "fov"),
    (dict(origin=[1, 1, 0], look_at=[0, 0, 0], aspect_ratio=0), "aspect"),
    (dict(origin=[1, 1, 0], look_at=[0, 0, 0], focus_dist=-2), "focus"),
    (dict(origin=None, look_at=[0, 0, 0]), "origin wasn't provided"),
])
def test_camera_builder_verify_errors(api, kw, msg):  # screen.rs:150-198
    with pytest.raises(api.RtError) as e:
        api.camera_build(**kw)
    assert e.value.code == api.RT_ERR_BAD_ARG and msg in str(e.value)


def test_birds_eye_view_up(api):  # view_up_degrees(15, Axis::Y), screen.rs:233-244
    w = api.World("BirdsEyeView", seed=1)
    rads = np.radians(15) + np.pi / 2
    up = np.array([np.cos(rads), 0, -np.sin(rads)])
    wv = np.array([0, 1.0, 0])
    u = np.cross(up, wv)
    u /= np.linalg.norm(u)
    assert np.allclose(w.camera.u[:], u, atol=1e-6)


# ---- the C ABI -----------------------------------------------------------------------------------------------
def _declared_functions():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if h.endswith(".h"))  # every header under include/
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(api):
    lib = api.load()
    names = _declared_functions()
    assert len(names) >= 20 and set(names) == set(api.ABI_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), f"librtx_hip.so does not export {n}"
    assert lib.rt_abi_version() == 1


def test_struct_layouts_match_the_header(api, tmp_path):
    """ctypes mirrors == what a C compiler lays out for include/rtx_hip.h."""
    names = ["rt_noise_table", "rt_texture", "rt_material", "rt_scene", "rt_camera", "rt_params", "rt_stats",
             "rt_ray", "rt_hit", "rt_scatter", "rt_multi_stats"]
    src = '#include <stdio.h>\n#include "rtx_hip.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + 'printf("%zu %zu\\n", __builtin_offsetof(rt_params, seed), __builtin_offsetof(rt_params, t_min)); return 0;}'
    c = tmp_path / "sz.c"
    c.write_text(src)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    mirrors = [api.NoiseTable, api.Texture, api.Material, api.Scene, api.Camera, api.Params, api.Stats]
    for m, sz in zip(mirrors, out[:7]):
        assert C.sizeof(m) == int(sz), m
    assert [api.ray_dtype.itemsize, api.hit_dtype.itemsize, api.scatter_dtype.itemsize] == [int(x) for x in out[7:10]]
    assert C.sizeof(api.MultiStats) == int(out[10])
    assert (api.Params.seed.offset, api.Params.t_min.offset) == (int(out[11]), int(out[12]))


def test_no_device_is_a_loud_error(api):
    """No CPU fallback: without a gfx950 device rt_create fails with RT_ERR_NO_DEVICE (on the GPU box it works)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtError) as e:
        api.Context(0)
    assert e.value.code in (api.RT_ERR_NO_DEVICE, api.RT_ERR_HIP)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value) or "device" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """The product path must not import / link / call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle's copy", "").replace("the oracle", "").replace(
                    "oracle/rt_oracle.hpp", "").replace("as the oracle", "").replace("oracle;", ""), (dirpath, f)
    out = subprocess.run(["ldd", os.path.join(ROOT, "raytracing_amd", "librtx_hip.so")], capture_output=True,
                         text=True).stdout
    assert "liboracle" not in out
