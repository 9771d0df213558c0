"""Quick end-to-end GPU sanity run (development aid, not a test): RNG parity, a ray batch vs the
oracle, a small render vs the oracle, and a first timing of the C2 workload at reduced spp."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as orc  # noqa: E402
from raytracing_amd import api  # noqa: E402


def main():
    ctx = api.Context(0)
    w = api.World("FinalStatic", seed=1, aspect=400 / 225)
    ctx.upload(w.scene)
    # RNG
    c = np.random.default_rng(0).integers(0, 2**32, (1000, 4), dtype=np.uint32)
    print("philox equal:", np.array_equal(ctx.rng_batch(12345678901234567, c), orc.philox(12345678901234567, c)))
    # camera rays -> intersect
    rng = np.random.default_rng(1)
    n = 20000
    ij = rng.random((n, 2), dtype=np.float32)
    u = rng.random((n, 4), dtype=np.float32)
    rays = ctx.camera_rays(w.camera, ij, u)
    ro = orc.camera_rays(w.camera, ij, u)
    print("camera dir max abs err", np.abs(rays["dir"] - ro["dir"]).max())
    t0 = time.time()
    hg = ctx.intersect_batch(rays)
    ho = orc.intersect_batch(w.scene, rays)
    ok = ho["margin"] >= 1.0
    print("primary: n", n, "excluded", (~ok).sum(), "hit mismatch", (hg["hit"][ok] != ho["hit"][ok]).sum(),
          "id mismatch", (hg["sphere_id"][ok] != ho["sphere_id"][ok]).sum())
    both = ok & (hg["hit"] == 1) & (ho["hit"] == 1) & (hg["sphere_id"] == ho["sphere_id"])
    print("t rel err max", (np.abs(hg["t"][both] - ho["t"][both]) / ho["t"][both]).max(),
          "normal err max", np.abs(hg["normal"][both] - ho["normal"][both]).max())
    # scatter
    sg = ctx.scatter_batch(rays, hg, u)
    so = orc.scatter_batch(w.scene, rays, hg, u)
    m = hg["hit"] == 1
    print("scatter dir abs err max", np.abs(sg["dir"][m] - so["dir"][m]).max(), "albedo err",
          np.abs(sg["albedo"][m] - so["albedo"][m]).max())
    # second-bounce rays
    rays2 = np.zeros(m.sum(), api.ray_dtype)
    rays2["origin"] = sg["origin"][m]
    rays2["dir"] = sg["dir"][m]
    hg2 = ctx.intersect_batch(rays2)
    ho2 = orc.intersect_batch(w.scene, rays2)
    ok2 = ho2["margin"] >= 1.0
    print("secondary: n", len(rays2), "excluded", (~ok2).sum(), "hit mismatch",
          (hg2["hit"][ok2] != ho2["hit"][ok2]).sum(), "id mismatch",
          (hg2["sphere_id"][ok2] != ho2["sphere_id"][ok2]).sum(), "hits", hg2["hit"].sum())
    both = ok2 & (hg2["hit"] == 1) & (ho2["hit"] == 1) & (hg2["sphere_id"] == ho2["sphere_id"])
    print("t rel err max", (np.abs(hg2["t"][both] - ho2["t"][both]) / ho2["t"][both]).max())
    # small render
    p = api.make_params(200, 112, 16, seed=w.render_seed)
    img, st = ctx.render(w.camera, p)
    ref, sec, rays_o = orc.render(w.scene, w.camera, p, mode=0, bvh_first=w.bvh_first, n_threads=16)
    g = lambda a: np.sqrt(np.clip(a, 0, 1))
    print("render 200x112x16: gpu rays", st.rays_traced, "oracle rays", rays_o, "kernel_ms", st.kernel_ms,
          "rmse(gamma)", np.sqrt(((g(img) - g(ref)) ** 2).mean()), "max abs", np.abs(img - ref).max())
    print("nan/inf", (~np.isfinite(img)).sum(), "min", img.min(), "max", img.max())
    # timing at C2 geometry, reduced spp
    w2 = api.World("FinalStatic", seed=1, aspect=1200 / 800)
    ctx.upload(w2.scene)
    for spp in (8, 64):
        p = api.make_params(1200, 800, spp, seed=w2.render_seed)
        st = ctx.render_accum_device_host(w2.camera, p) if hasattr(ctx, "render_accum_device_host") else ctx.render(w2.camera, p)[1]
        ns = 488 - 1
        flops = st.rays_traced * 17.0 * (w2.scene.n_spheres)
        print(f"1200x800x{spp}: kernel_ms {st.kernel_ms:.2f} total_ms {st.total_ms:.2f} rays {st.rays_traced} "
              f"rays/sample {st.rays_traced / st.samples:.3f} Msamples/s {st.samples / st.kernel_ms / 1e3:.1f} "
              f"Mrays/s {st.rays_traced / st.kernel_ms / 1e3:.1f} TFLOP/s(alg) {flops / st.kernel_ms / 1e9:.2f}")


if __name__ == "__main__":
    main()
