"""Debug: scatter_batch of two library builds on the golden batches, element by element."""
import os, subprocess, sys
import numpy as np

if len(sys.argv) == 3:
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, lib, "child", lib], capture_output=True, text=True)
        print(r.stdout[-1000:], r.stderr[-500:])
    for pre in ("scatter", "scatter_in", "scatter_gz"):
        a, b = (np.load(f"/tmp/abs_{os.path.basename(l)}_{pre}.npy") for l in sys.argv[1:])
        mt = np.load(f"/tmp/abs_mat_{pre}.npy")
        d = np.abs(a["dir"] - b["dir"]).max(1) + np.abs(a["albedo"] - b["albedo"]).max(1) + np.abs(a["origin"] - b["origin"]).max(1)
        bad = np.nonzero(d > 0)[0]
        print(pre, "differing", len(bad), "of", len(d), "material types of those", np.bincount(mt[bad], minlength=3), "index parity", np.bincount(bad % 2, minlength=2))
        for i in bad[:5]:
            print("  ", i, "type", mt[i], a["dir"][i], b["dir"][i], a["albedo"][i], b["albedo"][i])
else:
    os.environ["RTX_HIP_LIB"] = os.path.join(os.path.dirname(__file__), "..", "raytracing_amd", sys.argv[1])
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from raytracing_amd import api
    api.allow_lib_override()  # A/B tooling: RTX_HIP_LIB names the build under test
    ctx = api.Context(0)
    w = api.World("FinalStatic", seed=1, aspect=1.5)
    ctx.upload(w.scene)
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rays_FinalStatic.npz"))
    s = w.scene
    mats = np.array([s.materials[i].type for i in range(s.n_materials)])
    for pre in ("scatter", "scatter_in", "scatter_gz"):
        rays, hits, u = g[f"{pre}_rays"], g[f"{pre}_hits"], g[f"{pre}_u"]
        out = ctx.scatter_batch(rays, hits, u)
        np.save(f"/tmp/abs_{os.path.basename(sys.argv[1])}_{pre}.npy", out)
        np.save(f"/tmp/abs_mat_{pre}.npy", np.where(hits["hit"] == 1, mats[hits["material_id"]], 0))
