"""Shared helpers for the parity tests."""
import hashlib

import numpy as np


def scene_checksum(api, world) -> str:
    """Must match tests/golden/make_golden.py::scene_checksum."""
    C = api.C
    s = world.scene
    h = hashlib.sha256()
    a = world.arrays()
    for k in ("c0", "dc", "radius", "material_id"):
        h.update(np.ascontiguousarray(a[k]).tobytes())
    if s.n_materials:
        h.update(bytes(memoryview((api.Material * s.n_materials).from_address(C.addressof(s.materials.contents)))))
    if s.n_textures:
        h.update(bytes(memoryview((api.Texture * s.n_textures).from_address(C.addressof(s.textures.contents)))))
    if s.n_noise_tables:
        h.update(bytes(memoryview((api.NoiseTable * s.n_noise_tables).from_address(
            C.addressof(s.noise_tables.contents)))))
    h.update(bytes(memoryview(world.camera)))
    return h.hexdigest()


def rel_err(a, b, floor=1e-30):
    """|a-b| / max(|b|, floor), vector-wise over the last axis if 2-D."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.ndim == 2:
        return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), floor)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def check_hits(got, exp, eps_margin, tol=1e-5, max_excluded=0.001, margin="margin_tie"):
    """north_star check (1): hit/miss, sphere id, material id and front_face BIT-EXACT for every ray whose
    f64 decision margin exceeds eps_margin; t and normal within `tol` relative. Returns stats dict.
    margin = "margin_tie" (default): only rays with a root within 1e-6 (relative) of t_min or two spheres within 1e-9
    (relative) in depth are exempt -- last-bit differences between two f64 evaluations; every decision is made in
    f64 on the device (hit/miss, the t_min boundary, which sphere is closer, exact ties by world order), so there is
    no fp32 band of any kind.  max_excluded: at most that fraction of a batch may be exempt (default 0.1 %).
    margin = "margin": round 1's bar, which also exempted 16 fp32 ulps around a grazing discriminant."""
    ok = exp[margin] >= eps_margin
    n = len(exp)
    stats = {"n": n, "excluded": int((~ok).sum())}
    assert stats["excluded"] <= max_excluded * n, f"too many rays excluded by the margin: {stats}"
    for k in ("hit", "sphere_id", "material_id", "front_face"):
        bad = np.nonzero(ok & (got[k] != exp[k]))[0]
        assert len(bad) == 0, f"{k} differs on {len(bad)} of {ok.sum()} rays, first {bad[:5]}: got {got[k][bad[:5]]} exp {exp[k][bad[:5]]} margin {exp['margin'][bad[:5]]}"
    both = ok & (exp["hit"] == 1)
    if both.any():
        et = rel_err(got["t"][both], exp["t"][both])
        en = rel_err(got["normal"][both], exp["normal"][both])
        ep = np.linalg.norm(got["point"][both].astype(np.float64) - exp["point"][both], axis=1) / np.maximum(
            np.linalg.norm(exp["point"][both], axis=1), 1.0)
        stats.update(t_rel=float(et.max()), n_rel=float(en.max()), p_rel=float(ep.max()))
        assert et.max() <= tol, stats
        assert en.max() <= tol, stats
        assert ep.max() <= tol, stats
    return stats


def check_image(label, img, ref, tol, rays=None, rays_ref=None, rays_tol=None):
    """RMSE after gamma of a device render against the oracle's render of the same counter streams, and the ray counts.
    Prints what it measured (pytest -s) so that tolerances can be kept within a few times the measured value."""
    a = np.sqrt(np.clip(np.asarray(img, np.float64), 0, 1))
    b = np.sqrt(np.clip(np.asarray(ref, np.float64), 0, 1))
    r = float(np.sqrt(((a - b) ** 2).mean()))
    msg = f"{label}: rmse(gamma) {r:.3e} (tol {tol:.1e})"
    if rays is not None:
        rel = abs(int(rays) - int(rays_ref)) / max(int(rays_ref), 1)
        msg += f" rays {int(rays)} vs {int(rays_ref)} rel {rel:.2e} (tol {rays_tol:.1e})"
    print(msg)
    assert r < tol, msg
    if rays is not None:
        assert abs(int(rays) - int(rays_ref)) <= max(2, rays_tol * int(rays_ref)), msg
    return r
