"""Debug: trigger statistics of the lazy-root scan (needs librtx_stats.so built with -DRT_STATS)."""
import ctypes as C, os, sys
os.environ["RTX_HIP_LIB"] = os.path.join(os.path.dirname(__file__), "..", "raytracing_amd", os.environ.get("STATS_LIB", "librtx_stats.so"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from raytracing_amd import api
api.allow_lib_override()  # A/B tooling: RTX_HIP_LIB names the build under test
ctx = api.Context(0)
for scene in sys.argv[1:] or ["FinalStatic"]:
    w = api.World(scene, seed=1, aspect=1.5)
    ctx.upload(w.scene)
    out = (C.c_ulonglong * 8)()
    ctx.lib.rt_debug_counters(out, 1)
    spp = int(os.environ.get("SPP", "20"))
    p = api.make_params(1200, 800, spp, seed=w.render_seed)
    img, st = ctx.render(w.camera, p)
    ctx.lib.rt_debug_counters(out, 1)
    g, gt, stk, r_useful, r_front, s_front, s_front_bound = [out[i] for i in range(7)]
    # 0 member groups tested, 1 groups sent to root extraction, 2 spheres there, 3 ray-roots with disc >= 0,
    # 5 bounds tested (all levels), 6 group bounds passed
    print("raw", [int(out[i]) for i in range(8)], "rays", int(st.rays_traced))
    print(scene, "unfinished (kRetry) segment-passes per ray segment: %.5f" % (out[7] / st.rays_traced))
    print("slot utilisation %.4f (rays %d, passes %d, kernel %.1f ms)" % (st.rays_traced / (g / (w.arrays()["radius"].shape[0] // 4 + 0.0) * 128), st.rays_traced, g, st.kernel_ms))
    print(scene, "groups", g, "taken %.3f" % (gt / g), "spheres taken / sphere %.3f" % (stk / (4 * g)),
          "per taken group %.2f" % (stk / max(gt, 1)), "useful ray-roots per taken sphere (of 128) %.2f" % (r_useful / max(stk, 1)),
          "in-front %.2f" % (r_front / max(stk, 1)), "spheres w/ in-front hit / sphere %.3f" % (s_front / (4 * g)),
          "and within bound+1 %.3f" % (s_front_bound / (4 * g)))
    print(scene, "static groups whose bounding sphere some lane's ray meets: %.3f" % (r_front / g))
