"""Debug: fixed cost of a render launch (tiny workloads)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracing_amd import api
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
acc = torch.zeros(800, 1200, 3, dtype=torch.float32, device="cuda")
for (W, H, spp) in [(2, 2, 1), (1200, 800, 1), (1200, 800, 8), (1200, 800, 16), (1200, 800, 32), (1200, 800, 64)]:
    p = api.make_params(W, H, spp, seed=w.render_seed)
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    print(W, H, "spp", spp, "kernel_ms %.3f" % st.kernel_ms, "total_ms %.3f" % st.total_ms, flush=True)
