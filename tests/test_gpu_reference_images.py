"""north_star check (2), DIRECT: the converged device image against the REFERENCE's own sampling.

tests/golden/reference_images.npz (generator: tests/golden/make_reference_images.py) holds, for FinalStatic, Random
and PerlinTurbulence at configs[0] geometry (400x225), two INDEPENDENT oracle renders A and B at 1000 and at 4096
spp made the way the reference renders (oracle MODE_REFERENCE: f64, BVH, per-row sequential Pcg64Mcg seeded
(seed+1)*y as main.rs:71, rejection samplers as vec3.rs:53-62, the Dielectric uniform only when not TIR as
material.rs:101-102).  The device renders the same scene with ITS sampling (Philox counters, direct maps, fp32 scan).
The two share no random numbers, so the comparison is statistical, and every threshold below is DERIVED from the
A-vs-B pair, not hand-picked:

  floor(n)  = rmse_gamma(A_n, B_n)               Monte-Carlo noise of two unbiased renders at n spp each
  s^2       = floor(n)^2 * n / 2                 mean per-sample variance after gamma (floor ~ sqrt(2 s^2 / n))
  expect(n, m) = sqrt(s^2 (1/n + 1/m))           rmse of an unbiased m-spp render against an n-spp one

A device render with a systematic error e measures sqrt(expect^2 + e^2); SLACK = 3 % on the rmse leaves room for the
statistic's own scatter (it is an average over 270 000 values: its relative scatter is ~0.5 %) and bounds e at
0.25 * expect: 1.2e-3 (FinalStatic) at 1000 spp, 3e-4 for the 65536-spp render against the pooled 8192 reference
samples -- about a tenth of one 8-bit grey level.
"""
import numpy as np
import pytest

from conftest import golden
from helpers import scene_checksum

SCENES = ("FinalStatic", "Random", "PerlinTurbulence")
SLACK = 1.03
MEAN_TOL = 1e-4  # per-channel mean linear radiance (VERDICT asked for 1e-3; A and B agree to 4e-5, device measured 1.2e-5)


def _gamma(q):
    return q.astype(np.float64) / 65535.0


def _rmse(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()))


def test_fixture_is_self_consistent(api):
    """CPU: the scenes the fixture was rendered from are the ones the factory builds today, the noise floor scales
    as 1/sqrt(spp) (the renders are converging to ONE image), and A / B agree on the mean radiance."""
    g = golden("reference_images.npz")
    W, H = int(g["width"]), int(g["height"])
    for name in SCENES:
        w = api.World(name, seed=1, aspect=W / H)
        assert scene_checksum(api, w) == str(g[f"{name}_checksum"])
        f1 = _rmse(_gamma(g[f"{name}_1000_a"]), _gamma(g[f"{name}_1000_b"]))
        f4 = _rmse(_gamma(g[f"{name}_4096_a"]), _gamma(g[f"{name}_4096_b"]))
        assert abs(f1 / f4 / np.sqrt(4.096) - 1) < 0.02, (name, f1, f4)
        for spp in (1000, 4096):
            assert np.abs(g[f"{name}_{spp}_a_mean"] - g[f"{name}_{spp}_b_mean"]).max() < 1e-4
        # the quantised image reproduces the stored f64 mean radiance
        lin = _gamma(g[f"{name}_4096_a"]) ** 2
        assert np.abs(lin.mean((0, 1)) - g[f"{name}_4096_a_mean"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_converged_image_matches_the_reference_sampler_render(ctx, api, name):
    g = golden("reference_images.npz")
    W, H = int(g["width"]), int(g["height"])
    w = api.World(name, seed=1, aspect=W / H)
    assert scene_checksum(api, w) == str(g[f"{name}_checksum"])
    ctx.upload(w.scene)
    s2 = None
    for spp in (1000, 4096):
        A, B = _gamma(g[f"{name}_{spp}_a"]), _gamma(g[f"{name}_{spp}_b"])
        floor = _rmse(A, B)
        s2 = floor * floor * spp / 2
        img, st = ctx.render(w.camera, api.make_params(W, H, spp, seed=w.render_seed))
        assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= 1
        G = np.sqrt(img.astype(np.float64))
        ra, rb = _rmse(G, A), _rmse(G, B)
        rays_ref = 0.5 * (int(g[f"{name}_{spp}_a_rays"]) + int(g[f"{name}_{spp}_b_rays"]))
        rays_rel = abs(int(st.rays_traced) - rays_ref) / rays_ref
        print(f"{name} {spp} spp: noise floor rmse(A,B) {floor:.6f}  device vs A {ra:.6f} vs B {rb:.6f}  "
              f"rays/sample device {st.rays_traced / st.samples:.4f} ref {rays_ref / st.samples:.4f}")
        assert ra <= SLACK * floor and rb <= SLACK * floor, (name, spp, ra, rb, floor)
        # mean path length agrees with the reference's (the two differ only by Monte-Carlo scatter)
        assert rays_rel < 2e-3, (name, spp, rays_rel)
    # the sharp one: 65536 device spp against the 8192 pooled reference samples (A and B averaged in LINEAR radiance)
    ref_lin = 0.5 * (_gamma(g[f"{name}_4096_a"]) ** 2 + _gamma(g[f"{name}_4096_b"]) ** 2)
    img, st = ctx.render(w.camera, api.make_params(W, H, 65536, seed=w.render_seed))
    expect = float(np.sqrt(s2 * (1 / 8192 + 1 / 65536)))
    r = _rmse(np.sqrt(img.astype(np.float64)), np.sqrt(ref_lin))
    mean_ref = 0.5 * (g[f"{name}_4096_a_mean"] + g[f"{name}_4096_b_mean"])
    dmean = np.abs(img.astype(np.float64).mean((0, 1)) - mean_ref)
    print(f"{name} 65536 spp vs pooled 8192: rmse {r:.6f} expected {expect:.6f}; mean radiance diff {dmean}")
    assert r <= SLACK * expect, (name, r, expect)
    assert dmean.max() < MEAN_TOL, (name, dmean)
