#!/usr/bin/env python3
"""Build-time check of the hand-pinned scalar prefetch in scan_spheres (rt_device.h).

The next sphere group is fetched with inline-asm `s_load_dwordx16 / s_load_dwordx4 ... ; PF_LOAD` whose completion the
compiler does not track; the matching `s_waitcnt lgkmcnt(0) ; PF_WAIT` is the first point where the destination
SGPRs may be touched.  This script reads the kernel ISA (the Makefile emits rt_lib.s), builds the control-flow
graph of every function (labels, s_branch / s_cbranch_* targets, fall-through) and walks it forward from each
PF_LOAD: on every path, no instruction may name one of the destination SGPRs before a PF_WAIT is reached (a copy
or spill inserted by the register allocator, or a temporary allocated on top of them, would read or clobber them
while the data is still in flight), and no path may reach the end of the program without a PF_WAIT."""
import re
import sys


def sregs(text):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[:])s(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(path):
    raw = open(path).read().splitlines()
    # instruction list: (line number, code text, full text)
    ins, label_at = [], {}
    for no, ln in enumerate(raw, 1):
        code = ln.split(";")[0].strip()
        m = re.match(r"^([.\w$]+):", code)
        if m:
            label_at[m.group(1)] = len(ins)  # index of the next instruction
            continue
        if not code or code.startswith("."):
            if "PF_" not in ln:
                continue
        if code.startswith(";") or not code:
            continue
        ins.append((no, code, ln))
    n_loads, bad = 0, []

    def successors(i):
        code = ins[i][1]
        op = code.split()[0]
        if op == "s_endpgm":
            return []
        if op == "s_branch":
            t = code.split()[1]
            return [label_at[t]] if t in label_at else []
        if op.startswith("s_cbranch"):
            t = code.split()[-1]
            out = [i + 1] if i + 1 < len(ins) else []
            if t in label_at:
                out.append(label_at[t])
            return out
        if op in ("s_setpc_b64", "s_swappc_b64"):
            return [None]  # indirect: cannot be followed
        return [i + 1] if i + 1 < len(ins) else []

    for i, (no, code, full) in enumerate(ins):
        if "PF_LOAD" not in full or not code.startswith("s_load_dword"):
            continue
        n_loads += 1
        m = re.search(r"s_load_dword(?:x\d+)?\s+(?:s\[(\d+):(\d+)\]|s(\d+))", code)  # x16 sphere group, x4 -r^2 group
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
        seen, stack = set(), [j for j in successors(i)]
        while stack:
            j = stack.pop()
            if j is None:
                bad.append((no, no, "indirect branch between PF_LOAD and PF_WAIT"))
                continue
            if j in seen or j >= len(ins):
                continue
            seen.add(j)
            jno, jcode, jfull = ins[j]
            if "PF_WAIT" in jfull or (jcode.startswith("s_waitcnt") and "lgkmcnt(0)" in jcode):
                continue  # this path is safe from here on (ANY full lgkm wait means the data has arrived)
            if jcode.split()[0] == "s_endpgm":
                bad.append((no, jno, "program ends before PF_WAIT"))
                continue
            if "PF_LOAD" in jfull:  # the sibling load of a two-load asm (or the next prefetch): its own walk checks it
                hit = sregs(jcode.split(",")[0]) & dst
            else:
                hit = sregs(jcode) & dst
            if hit:
                bad.append((no, jno, "touches s%s before PF_WAIT: %s" % (sorted(hit), jcode)))
                continue
            stack.extend(successors(j))
    return n_loads, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "raytracing_amd/csrc/rt_lib.s"
    n, bad = check(path)
    if n == 0:  # nothing is pinned by hand in this build: the compiler tracks every scalar load itself
        print("check_prefetch: no PF_LOAD in", path, "(nothing to check)")
        sys.exit(0)
    for b in bad[:40]:
        print("check_prefetch: load at line %d, line %d: %s" % b)
    print("check_prefetch: %d prefetch loads, %d violations" % (n, len(bad)))
    sys.exit(1 if bad else 0)
