"""The tail-order experiment (rt_lib.hip: build_tail_order).  Needs a library built with -DRT_X_TAILORDER
(the Makefile's flags + the define, e.g. raytracing_amd/librtx_tailorder.so; RTX_HIP_LIB names it).
1. Scheduling only: renders with the order on and off (rt_debug_set_tail_order) are identical bit for bit, the order read
   back from the device is a permutation of the tiles, and tiles do get marked (small units, so that the launch has a
   history to learn from at these sizes; a path is counted after 32 passes, ~1.5 ms).
2. What it is worth: kernel time with the order on and off, per scene and sample count, full size."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("RTX_HIP_LIB", os.path.join(ROOT, "raytracing_amd", "librtx_tailorder.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raytracing_amd import api
api.allow_lib_override()
ctx = api.Context(0)
for scene, (wd, ht, spp), units in (("FinalStatic", (240, 160, 300), (0, 0, -1)), ("FinalStatic", (240, 160, 300), (8, 4, 64)),
                                    ("FinalStatic", (200, 120, 600), (8, 2, 100)), ("Random", (168, 104, 800), (8, 4, 64))):
    w = api.World(scene, seed=1, aspect=wd / ht)
    ctx.upload(w.scene)
    p = api.make_params(wd, ht, spp, seed=w.render_seed)
    res = {}
    for on in (True, False):
        ctx.debug_set_tail_order(on)
        ctx.debug_set_work_units(*units)
        res[on] = ctx.render(w.camera, p)
        if on:
            stats = ctx.debug_tail_order_stats()
    ctx.debug_set_tail_order(True)
    ctx.debug_set_work_units()
    n_tiles = ((wd + 7) // 8) * ((ht + 7) // 8)
    same = np.array_equal(res[True][0], res[False][0]) and res[True][1].rays_traced == res[False][1].rays_traced
    print("identity", scene, wd, ht, spp, units, stats, "of", n_tiles, "tiles:", "identical" if same else "DIFFERENT", flush=True)
    assert stats["built"] == 1 and stats["valid"] == 1 and same, "not an experiment build, or the order is broken"
    assert units[0] == 0 or 0 < stats["marked_tiles"] < n_tiles
for scene in sys.argv[1:] or ["FinalStatic", "Random"]:
    w = api.World(scene, seed=1, aspect=1.5)
    ctx.upload(w.scene)
    acc = torch.zeros(800, 1200, 3, dtype=torch.float32, device="cuda")
    for spp in [int(x) for x in os.environ.get("SPPS", "250,500,1000").split(",")]:
        p = api.make_params(1200, 800, spp, seed=w.render_seed)
        ms = {True: [], False: []}
        for on in (True, False, True, False, True, False):
            ctx.debug_set_tail_order(on)
            st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
            ms[on].append(st.kernel_ms)
            if on:
                stats = ctx.debug_tail_order_stats()
        print("timing", scene, "spp", spp, stats, "kernel ms with the order %s, without %s" % (["%.2f" % v for v in ms[True]], ["%.2f" % v for v in ms[False]]), flush=True)
