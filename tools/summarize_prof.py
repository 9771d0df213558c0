#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory (rocprofv3 rocpd .db files) into a small text summary
and hbm_traffic.json.  Usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag>"""
import json
import os
import sqlite3
import sys


def rows(db, q):
    if not os.path.exists(db):
        return []
    try:
        return list(sqlite3.connect(db).execute(q))
    except Exception as e:  # noqa: BLE001
        return [("error", str(e))]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = []
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        out.append("## bench.py line (same command, un-profiled)\n```\n" + open(bench).read().strip() + "\n```\n")
    out.append("## rocprofv3 --kernel-trace --stats (top kernels; durations in ns)\n")
    out.append("| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|")
    for r in rows(os.path.join(src, "stats", "stats_results.db"), "select * from top_kernels"):
        out.append("| " + " | ".join(str(x)[:90] for x in r) + " |")
    counters = {}
    out.append("\n## rocprofv3 --pmc passes (separate runs; per-kernel SUM over dispatches, and dispatch count)\n")
    out.append("| pass | kernel | counter | dispatches | sum | per dispatch |\n|---|---|---|---|---|---|")
    for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_lds"):
        q = ("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
             "group by kernel_name, counter_name")
        for k, c, n, s in rows(os.path.join(src, p, "pmc_results.db"), q):
            if "render_kernel" in str(k) or "scan" in str(k):
                out.append(f"| {p} | {str(k)[:60]} | {c} | {n} | {s:.6g} | {s / n:.6g} |")
                counters[c] = s / n
    derived = []
    if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE reads 1/2 of streamed bytes
        # (MI355X_MICROARCH.md §HBM) -> double it.
        fetch = counters.get("FETCH_SIZE", 0.0) * 1024 * 2
        write = counters.get("WRITE_SIZE", 0.0) * 1024
        derived.append(f"HBM bytes per render_kernel launch: fetch {fetch:.4g} (x2-corrected) + write {write:.4g} = {fetch + write:.4g}")
        # one entry per measured workload; bench.py only quotes the entry whose config matches its arguments
        tpath = os.path.join(os.path.dirname(dst), "hbm_traffic.json")
        try:
            entries = json.load(open(tpath)).get("entries", [])
        except Exception:
            entries = []
        cfg = {}
        if os.path.exists(bench):
            c = json.loads(open(bench).read().strip().splitlines()[-1])["config"]
            cfg = {"scene": c["scene"], "width": c["width"], "height": c["height"], "spp": c["spp_per_gpu"]}
        entries = [e for e in entries if e.get("config") != cfg]
        entries.append({"config": cfg, "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                        "source": os.path.basename(src)})
        json.dump({"entries": entries}, open(tpath, "w"), indent=1)
    if "SQ_INSTS_VALU" in counters and "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters:
        # one entry per measured workload (like hbm_traffic.json): bench.py quotes the executed-work figures of the
        # entry whose config matches its arguments
        vpath = os.path.join(os.path.dirname(dst), "valu_pmc.json")
        try:
            ventries = json.load(open(vpath)).get("entries", [])
        except Exception:
            ventries = []
        cfg = {}
        if os.path.exists(bench):
            c = json.loads(open(bench).read().strip().splitlines()[-1])["config"]
            cfg = {"scene": c["scene"], "width": c["width"], "height": c["height"], "spp": c["spp_per_gpu"]}
        ventries = [e for e in ventries if e.get("config") != cfg]
        ventries.append({"config": cfg, "source": os.path.basename(src),
                         "valu_insts_per_launch": counters["SQ_INSTS_VALU"],
                         "active_inst_valu": counters["SQ_ACTIVE_INST_VALU"],
                         "thread_cycles_valu": counters["SQ_THREAD_CYCLES_VALU"],
                         "salu_insts_per_launch": counters.get("SQ_INSTS_SALU"),
                         "smem_insts_per_launch": counters.get("SQ_INSTS_SMEM"),
                         "waves": counters.get("SQ_WAVES")})
        json.dump({"entries": ventries}, open(vpath, "w"), indent=1)
    if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters:
        derived.append("VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)): "
                       f"{counters['SQ_THREAD_CYCLES_VALU'] / (64 * counters['SQ_ACTIVE_INST_VALU']):.3f}")
    if "GRBM_GUI_ACTIVE" in counters:
        derived.append(f"GRBM_GUI_ACTIVE / 8 XCDs = {counters['GRBM_GUI_ACTIVE'] / 8:.4g} shader cycles per launch")
    if "SQ_INSTS_VALU" in counters:
        derived.append(f"SQ_INSTS_VALU per launch: {counters['SQ_INSTS_VALU']:.4g} wave-instructions")
    out.append("\n## derived\n" + "\n".join("- " + d for d in derived))
    open(dst + "_summary.md", "w").write("\n".join(out) + "\n")
    print("\n".join(out[-8:]))


if __name__ == "__main__":
    main()
