"""Debug: kernel time vs spp on configs[1] geometry (fixed cost / tail vs per-sample cost)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracing_amd import api
if os.environ.get('RTX_HIP_LIB'):
    api.allow_lib_override()  # A/B tooling (tools/spp_sweep_libs.sh)
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
acc = torch.zeros(800, 1200, 3, dtype=torch.float32, device="cuda")
for spp in [int(x) for x in os.environ.get('SPPS', '125,250,500,1000,2000').split(',')]:
    p = api.make_params(1200, 800, spp, seed=w.render_seed, max_depth=int(os.environ.get("MAXD", "100")))
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    st = ctx.render_accum_device(w.camera, p, acc.data_ptr())
    print("spp", spp, "kernel_ms %.2f" % st.kernel_ms, "ms per 1000 spp %.2f" % (st.kernel_ms * 1000 / spp), flush=True)
