#!/usr/bin/env python3
"""Build-time check of the hand-pinned scalar prefetch in scan_spheres (rt_device.h).

The next sphere group is fetched with an inline-asm `s_load_dwordx16 ... ; PF_LOAD` whose completion the
compiler does not track; the matching `s_waitcnt lgkmcnt(0) ; PF_WAIT` is the first point where the destination
SGPRs may be read.  This script reads the kernel ISA (make -C raytracing_amd/csrc asm -> rt_lib.s) and fails if
any instruction between a PF_LOAD and the next PF_WAIT names one of the destination SGPRs (a copy or spill the
register allocator might have inserted would read them before the data has arrived)."""
import re
import sys


def sregs(text):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(path):
    lines = open(path).read().splitlines()
    n_loads, bad = 0, []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if "PF_LOAD" in ln and "s_load_dwordx16" in ln:
            n_loads += 1
            m = re.search(r"s_load_dwordx16\s+s\[(\d+):(\d+)\]", ln)
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            j = i + 1
            while j < len(lines) and "PF_WAIT" not in lines[j]:
                code = lines[j].split(";")[0]
                if code.strip() and not code.strip().endswith(":") and not code.strip().startswith("."):
                    if "s_endpgm" in code:
                        bad.append((i + 1, j + 1, "kernel ends before PF_WAIT"))
                        break
                    hit = sregs(code) & dst
                    if hit:
                        bad.append((i + 1, j + 1, "reads/writes s%s before PF_WAIT: %s" % (sorted(hit), code.strip())))
                j += 1
            if j == len(lines):
                bad.append((i + 1, j, "no PF_WAIT after PF_LOAD"))
        i += 1
    return n_loads, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "raytracing_amd/csrc/rt_lib.s"
    n, bad = check(path)
    if n == 0:
        print("check_prefetch: no PF_LOAD found in", path)
        sys.exit(1)
    for b in bad:
        print("check_prefetch: line %d..%d: %s" % b)
    print("check_prefetch: %d prefetch loads, %d violations" % (n, len(bad)))
    sys.exit(1 if bad else 0)
