"""Ray regrouping (second half; hit.rs:264-283), simulated on the CPU before anything is built: how
many group bounds would a wave's 128 rays meet if the 512 in-flight rays of a workgroup (four waves, each on its own
8x8 tile x 48 spp unit) were re-dealt every pass by direction octant (and origin cell), against today's dealing (a
slot keeps its path)?  Paths come from the oracle (camera rays, hits, scatters with random uniforms) run through the
kernel's slot machine (128 slots per wave, a dead slot takes the unit's next item).  Bounds are the group bounds of a
Morton-ordered list of the scene's small spheres, groups of 4 and runs of 16, as upload builds them; "met" is the
kernel's test: the ray's line within R of the centre and the bound not wholly behind the ray.
Prints pass rates and a VALU-cost model per pass (planar bound 10, member group 29, +6 per regrouped slot)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import api
from oracle import binding as orc

scene = sys.argv[1] if len(sys.argv) > 1 else "FinalStatic"
W, H, SPP = 1200, 800, 48
w = api.World(scene, seed=1, aspect=W / H)
A = w.arrays()
rad, c0 = A["radius"].astype(np.float64), A["c0"].astype(np.float64)
small = np.nonzero(rad <= 32 * np.sort(rad)[(len(rad) - 1) // 2])[0]

def morton(c):
    lo = c.min(0); ext = max((c.max(0) - lo).max(), 1e-30)
    q = np.minimum(((c - lo) / ext * 2097151.0), 2097151.0).astype(np.uint64)
    code = np.zeros(len(c), np.uint64)
    for b in range(21):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(code, kind="stable")

order = small[morton(c0[small])]
def bounds(span):
    out = []
    for k in range(0, len(order), span):
        ids = order[k:k + span]
        ctr = c0[ids].mean(0)
        R = (np.linalg.norm(c0[ids] - ctr, axis=1) + rad[ids]).max()
        out.append((ctr, R))
    return np.array([b[0] for b in out]), np.array([b[1] for b in out])
gC, gR = bounds(4)
cC, cR = bounds(16)

def met(o, d, C, R):
    """(rays, bounds) -> bool: the kernel's bound test (line within R, not wholly behind)"""
    dh = d / np.linalg.norm(d, axis=1, keepdims=True)
    oc = C[None, :, :] - o[:, None, :]
    b = (oc * dh[:, None, :]).sum(2)
    q = (oc * oc).sum(2) - R[None, :] ** 2
    return (b * b - q >= 0) & ~((b < 0) & (q > 0))

rng = np.random.default_rng(1)
def unit_items(tx, ty):
    ys, xs = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    px = np.tile(np.stack([tx * 8 + xs.ravel(), ty * 8 + ys.ravel()], 1), (SPP, 1)).astype(np.float64)
    ij = np.stack([(px[:, 0] + rng.random(len(px))) / (W - 1), 1 - (px[:, 1] + rng.random(len(px))) / (H - 1)], 1).astype(np.float32)
    return orc.camera_rays(w.camera, ij, rng.random((len(px), 4), dtype=np.float32))

def to_f32(r):
    o = np.zeros(len(r), api.ray_dtype)
    o["origin"], o["dir"], o["time"] = r["origin"], r["dir"], r["time"]
    return o

NW = int(os.environ.get("WAVES", "4"))  # waves whose rays are pooled (4 = one workgroup; 16 / 64 = a wavefront queue across workgroups)
tot = {"now": [0, 0, 0], "oct": [0, 0, 0], "oct+cell": [0, 0, 0], "dir64": [0, 0, 0]}  # group bounds met, chunk bounds met, groups under met chunks
n_wave_pass = 0
for blk in range(int(os.environ.get("BLOCKS", "6"))):
    tx0, ty0 = int(rng.integers(10, 136)), int(rng.integers(30, 95))
    queues = [to_f32(unit_items(tx0 + k % 8, ty0 + k // 8)) for k in range(NW)]  # four neighbouring tiles: consecutive units
    nxt = [0] * NW
    slots = [np.zeros(128, api.ray_dtype) for _ in range(NW)]
    alive = [np.zeros(128, bool) for _ in range(NW)]
    depth = [np.zeros(128, int) for _ in range(NW)]
    for p in range(60):
        for k in range(NW):  # refill
            dead = np.nonzero(~alive[k])[0]
            take = min(len(dead), len(queues[k]) - nxt[k])
            slots[k][dead[:take]] = queues[k][nxt[k]:nxt[k] + take]
            alive[k][dead[:take]] = True; depth[k][dead[:take]] = 0
            nxt[k] += take
        if p >= 8:  # steady state: count
            allr = np.concatenate([slots[k][alive[k]] for k in range(NW)])
            o, d = allr["origin"].astype(np.float64), allr["dir"].astype(np.float64)
            octant = (d[:, 0] > 0) * 1 + (d[:, 1] > 0) * 2 + (d[:, 2] > 0) * 4
            cell = (o[:, 0] > np.median(o[:, 0])) * 1 + (o[:, 2] > np.median(o[:, 2])) * 2
            dh = d / np.linalg.norm(d, axis=1, keepdims=True)
            on = dh[:, :2] / np.abs(dh).sum(1, keepdims=True)  # octahedral map of the direction, 8 x 8 cells, Morton order
            on = np.where(dh[:, 2:3] < 0, (1 - np.abs(on[:, ::-1])) * np.sign(on + 1e-30), on)
            qx, qy = np.minimum(((on[:, 0] + 1) * 4).astype(int), 7), np.minimum(((on[:, 1] + 1) * 4).astype(int), 7)
            dir64 = sum((((qx >> b) & 1) << (2 * b)) | (((qy >> b) & 1) << (2 * b + 1)) for b in range(3))
            rank = lambda key: np.argsort(np.argsort(key, kind="stable"), kind="stable") * NW // len(allr)
            deals = {"now": np.concatenate([np.full(int(alive[k].sum()), k) for k in range(NW)]),
                     "oct": rank(octant), "oct+cell": rank(octant * 4 + cell), "dir64": rank(dir64)}
            mg, mc = met(o, d, gC, gR), met(o, d, cC, cR)
            for name, deal in deals.items():
                for k in range(NW):
                    sel = deal == k
                    if not sel.any():
                        continue
                    g_any, c_any = mg[sel].any(0), mc[sel].any(0)
                    tot[name][0] += int(g_any.sum()); tot[name][1] += int(c_any.sum())
                    tot[name][2] += int(np.repeat(c_any, 4)[:len(g_any)].sum())
            n_wave_pass += NW
        for k in range(NW):  # trace + scatter
            idx = np.nonzero(alive[k])[0]
            if len(idx) == 0:
                continue
            hits = orc.intersect_batch(w.scene, slots[k][idx])
            h32 = np.zeros(len(idx), api.hit_dtype)
            for f in ("hit", "sphere_id", "material_id", "front_face"):
                h32[f] = hits[f]
            h32["t"], h32["point"], h32["normal"] = hits["t"], hits["point"], hits["normal"]
            sc = orc.scatter_batch(w.scene, slots[k][idx], h32, rng.random((len(idx), 4), dtype=np.float32))
            hit = hits["hit"] == 1
            depth[k][idx] += 1
            cont = hit & (depth[k][idx] < 100)
            alive[k][idx[~cont]] = False
            s2 = slots[k][idx]
            s2["origin"][cont], s2["dir"][cont] = sc["origin"][cont], sc["dir"][cont]
            slots[k][idx] = s2
ng, nc = len(gR), len(cR)
print(f"{scene}: {ng} group bounds, {nc} chunk (16-sphere) bounds, {n_wave_pass} wave-passes")
for name, (g, c, gu) in tot.items():
    flat = ng * 10 + g / n_wave_pass * 29
    hier = nc * 10 + gu / n_wave_pass * 10 + g / n_wave_pass * 29
    extra = 0 if name == "now" else 128 * 6 / 64 * 2  # ~6 VALU per slot per pass for keys / addresses (LDS traffic aside)
    print(f"  {name:9s} group bounds met {g / n_wave_pass / ng:.3f}  chunk bounds met {c / n_wave_pass / nc:.3f}  "
          f"scan VALU/pass: flat walk {flat + extra:.0f}, with the 16-sphere level {hier + extra:.0f}")
