import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api
from oracle import binding as orc
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
W, H = 1200, 800
def R(spp, off, force=0):
    p = api.make_params(W, H, spp, seed=w.render_seed, spp_offset=off)
    p.reserved = force
    return ctx.render(w.camera, p)[0].astype(np.float64)
s0, s1 = R(1, 0), R(1, 1)
two = R(2, 0, force=2)   # sample 0 in slot A, sample 1 in slot B of the same lanes
b1 = two * 2 - s0        # sample 1 as rendered in slot B (if sample 0 is slot-independent)
d = np.abs(b1 - s1).max(2)
ys, xs = np.nonzero(d > 1e-4)
print("sample 1: slotB vs slotA pixels differing:", len(ys), list(zip(ys[:8].tolist(), xs[:8].tolist())))
# swap: sample 1 first => run spp=2 from offset 1 (samples 1,2): sample 1 in slot A with a live partner
s2 = R(1, 2)
two12 = R(2, 1, force=2)
d2 = np.abs(two12 * 2 - s1 - s2).max(2)
print("samples 1,2 paired vs singles:", int((d2 > 1e-4).sum()))
ref1 = orc.render(w.scene, w.camera, api.make_params(W, H, 1, seed=w.render_seed, spp_offset=1), mode=0, n_threads=16)[0]
for (y, x) in list(zip(ys[:8].tolist(), xs[:8].tolist())):
    print((y, x), "slotA", s1[y, x], "slotB", b1[y, x], "oracle", ref1[y, x])
