"""Debug: kernel time vs spp-per-chunk (rt_params.reserved debug hook) on configs[1]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracing_amd import api
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
acc = torch.zeros(800, 1200, 3, dtype=torch.float32, device="cuda")
for chunk in [0, 1000, 500, 250, 125, 63, 32]:
    p = api.make_params(1200, 800, 1000, seed=w.render_seed)
    ctx.debug_set_work_units(chunk, chunk, 0)
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    print("spp/chunk", chunk, "kernel_ms %.2f" % st.kernel_ms, "launches", st.n_launches, flush=True)
