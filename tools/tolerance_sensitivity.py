"""Do the converged-image tolerances of tests/test_gpu_parity.py::test_render_converged_rmse_after_gamma catch a build
that loses hits?  Renders the golden scenes with the library named by RTX_HIP_LIB (e.g. a -DRT_X_NORETRY build: a
rejected nominee is final, round 1's lost-hit bug) and prints rmse(gamma) against the committed oracle images next to
the test's tolerance."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import api
api.allow_lib_override()
from conftest import golden, rmse_gamma

TOL = {"FinalStatic": 6e-5, "Random": 1.5e-4, "PerlinMarbled": 5e-5, "PerlinTurbulence": 6e-5, "TwoSpheres": 2.5e-4}
ctx = api.Context(0)
g = golden("images.npz")
print("library", api.lib_provenance()["path"])
for name, tol in TOL.items():
    W, H, spp = (int(x) for x in g[f"{name}_shape"])
    w = api.World(name, seed=1, aspect=W / H)
    ctx.upload(w.scene)
    img, st = ctx.render(w.camera, api.make_params(W, H, spp, seed=int(g[f"{name}_seed"])))
    r = rmse_gamma(img, g[f"{name}_img"])
    print(f"{name:18s} rmse(gamma) {r:.3e}  tolerance {tol:.1e}  {'FAILS' if r >= tol else 'passes'}  rays {int(st.rays_traced)} vs {int(g[f'{name}_rays'])}")
