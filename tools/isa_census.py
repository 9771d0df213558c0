#!/usr/bin/env python3
"""Static instruction census of a render kernel by region (ISA built with -DRT_CENSUS -S --cuda-device-only: the
RT_MARK comments of render_kernel split the pass).  usage: isa_census.py file.s [kernel-substring]
Regions follow the ISA text order, so code the scheduler moved across a marker is counted with its new neighbours;
outside the scan the pass is straight-line code and the static count is what a wave issues per pass."""
import collections, re, sys

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "render_kernelILb0"
Q = re.compile(r"v_(sin|cos|rcp|rsq|sqrt|exp|log)_")
inside, cur, order = False, "prologue", []
cnt, w, f64, pk, mem, sal = (collections.Counter() for _ in range(6))
for line in open(path):
    t = line.strip()
    if re.match(r"_Z\w+:", t):
        inside, cur = kern in t, "prologue"
        continue
    if not inside:
        continue
    if t.startswith("; RT_MARK"):
        name = t.split()[2]
        cur = "%s#%d" % (name, sum(1 for o in order if o.split("#")[0] == name))
        order.append(cur)
        continue
    if t.startswith("s_endpgm"):
        inside = False
    op = t.split()[0] if t else ""
    if op.startswith("v_"):
        cnt[cur] += 1
        w[cur] += 4 if Q.match(op) else 1
        f64[cur] += "f64" in op
        pk[cur] += op.startswith("v_pk_")
    elif op.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_")):
        mem[cur] += 1
    elif op.startswith("s_"):
        sal[cur] += 1
tot = 0
for k in ["prologue"] + order:
    print(f"{k:14s} valu {cnt[k]:5d} (x4 for quarter rate: {w[k]:5d})  packed {pk[k]:4d} f64 {f64[k]:4d} mem {mem[k]:3d} salu {sal[k]:4d}")
    if not k.startswith(("scan", "prologue", "claim")):
        tot += w[k]
print("outside the scan, per pass (weighted):", tot)
