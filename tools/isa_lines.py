#!/usr/bin/env python3
"""Static VALU-instruction census of one kernel by source line (ISA built with -gline-tables-only -S).

usage: isa_lines.py file.s <kernel-name-substring> [top]
Outside the scan loops the render kernel's pass is straight-line code, so the static count of a line is what a
wave pays for it per pass; quarter-rate instructions (transcendentals, f64 rcp/rsq/sqrt) are weighted 4.
"""
import collections, re, sys

path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files, cur, inside = {}, None, False
count, weight = collections.Counter(), collections.Counter()
QUARTER = re.compile(r"v_(sin|cos|rcp|rsq|sqrt|exp|log)_")
for line in open(path):
    t = line.strip()
    m = re.match(r"\.file\s+(\d+)\s+(?:\"[^\"]*\"\s+)?\"([^\"]+)\"", t)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
        continue
    if re.match(r"_Z\w+:", t):
        inside = kern in t
        continue
    if not inside:
        continue
    if t.startswith(".loc"):
        p = t.split()
        cur = (files.get(int(p[1]), p[1]), int(p[2]))
        continue
    if t.startswith("s_endpgm"):
        inside = False
    if t.startswith("v_") and cur:
        op = t.split()[0]
        count[cur] += 1
        weight[cur] += 4 if QUARTER.match(op) else 1
print("total VALU", sum(count.values()), "weighted", sum(weight.values()))
for k, w in weight.most_common(top):
    print(f"{k[0]}:{k[1]:5d}  {count[k]:5d}  w {w:5d}")
