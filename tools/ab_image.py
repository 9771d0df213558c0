"""Debug: render the same small image with two library builds and compare (tools/ab_image.py libA.so libB.so)."""
import os, subprocess, sys
import numpy as np

if len(sys.argv) == 3:
    outs = []
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, lib, "child", lib], capture_output=True, text=True)
        print(r.stdout[-2000:], r.stderr[-500:])
    for depth in (1, 2, 3, 50):
        a, b = (np.load(f"/tmp/ab_{os.path.basename(l)}_{depth}.npy") for l in sys.argv[1:])
        d = np.abs(a - b)
        print("depth", depth, "pixels differing", int((d.max(2) > 0).sum()), "of", d.shape[0] * d.shape[1], "max", float(d.max()),
              "mean a", a.mean((0, 1)), "mean b", b.mean((0, 1)))
        if depth == 2:
            m = d.max(2) > 0
            for y in range(0, m.shape[0]):
                print("".join("#" if v else "." for v in m[y]))
else:
    os.environ["RTX_HIP_LIB"] = os.path.join(os.path.dirname(__file__), "..", "raytracing_amd", sys.argv[1])
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from raytracing_amd import api
    api.allow_lib_override()  # A/B tooling: RTX_HIP_LIB names the build under test
    ctx = api.Context(0)
    w = api.World(os.environ.get("SCENE", "FinalStatic"), seed=1, aspect=100 / 56)
    ctx.upload(w.scene)
    for depth in (1, 2, 3, 50):
        p = api.make_params(100, 56, 16, seed=w.render_seed, max_depth=depth)
        img, st = ctx.render(w.camera, p)
        print(sys.argv[1], "depth", depth, "rays", st.rays_traced)
        np.save(f"/tmp/ab_{os.path.basename(sys.argv[1])}_{depth}.npy", img)
