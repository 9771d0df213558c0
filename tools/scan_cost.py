"""Per-ray cost vs sphere count: fits kernel time = rays * (overhead + n_tests * per_test). Development aid."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api
ctx = api.Context(0)
w = api.World(sys.argv[1] if len(sys.argv) > 1 else "FinalStatic", seed=1, aspect=1.5)
full = w.scene.n_spheres
rows = []
for k in (1, 9, 33, 65, 129, 257, 385, full):
    s = w.scene
    s.n_spheres = k
    ctx.upload(s)
    p = api.make_params(1200, 800, 64, seed=3)
    ctx.debug_set_work_units(64, 64, 0)  # one chunk: 4096 items per wave, so the per-wave tail does not pollute the fit
    ctx.render(w.camera, p)
    _, st = ctx.render(w.camera, p)
    clk = 2.39e9
    cyc_per_ray_wave = st.kernel_ms * 1e-3 * clk * 1024 / (st.rays_traced / 128)  # per wave-pass: 64 lanes x 2 paths
    rows.append((k, cyc_per_ray_wave))
    print(f"n={k:4d} rays/sample {st.rays_traced/st.samples:.2f} kernel_ms {st.kernel_ms:8.2f} Grays/s {st.rays_traced/st.kernel_ms/1e6:6.2f} SIMD-cycles per wave-ray {cyc_per_ray_wave:9.0f}")
a = np.array(rows, float)
slope, icpt = np.polyfit(a[:, 0], a[:, 1], 1)
print(f"fit: {icpt:.0f} cycles overhead per wave-bounce + {slope:.1f} cycles per sphere test (wave64)")
