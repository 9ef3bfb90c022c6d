"""Condense rocprofv3 outputs into small tracked summaries.

    python tools/summarize_prof.py <tag> [--src gpurun_out/prof] [--out profiles]

Run ON THE GPU BOX at the end of tools/gpu_profile.sh (with --out gpurun_out/prof_summary: the raw kernel trace is too
large to travel) and copy the results into profiles/.  Writes
  <tag>_kernel_stats.csv     per (kernel, grid): calls, total/avg/min/max duration from the kernel trace — keyed by grid
                             because one kernel name serves several layer shapes (the 512->512 GRU and the 1024->128 one
                             are both pn_gru_mfma_p_kernel); roofline.frac is reproducible from this file alone
  <tag>_pmc_per_launch.csv   per (kernel, grid, counter): avg/min/max over the launches of the separate --pmc passes
Both files start with `# kernels_snapshot=<id>` (bench.kernels_snapshot(): sha256 over the kernel sources), so a reader
can tell which kernels were profiled; bench.py refuses PMC traffic from another snapshot.
"""
import argparse, collections, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "prof"))
ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
ap.add_argument("--suffix", default="", help="configuration suffix of the output names: '' (headline), _1024, _fp16, _x3")
ap.add_argument("--frames-total", type=int, default=0, help="frames the profiled bench ran (warm-up + steps); with --frames-keep: only the "
                "launches of the LAST frames-keep frames of every (kernel, grid) are summarised — the steady state.  The history ring holds 12 "
                "frames: the first frames of a run from the zero state filter at degenerate pitch periods (T = 768) and move more bytes")
ap.add_argument("--frames-keep", type=int, default=0)
a = ap.parse_args()
SRC, out, tag, suf = a.src, a.out, a.tag, a.suffix


def steady(rows):
    """rows: [(dispatch id, value)] of one (kernel, grid) -> the values of the last frames-keep frames (all, if not asked)"""
    rows = sorted(rows)
    if a.frames_total > 0 and 0 < a.frames_keep < a.frames_total and len(rows) >= a.frames_total and len(rows) % a.frames_total == 0:
        per_frame = len(rows) // a.frames_total
        rows = rows[-per_frame * a.frames_keep:]
    return [v for _, v in rows]

os.makedirs(out, exist_ok=True)
import bench
snap = bench.kernels_snapshot()


def short(name):
    n = name.replace("void ", "")
    if n.startswith("_Z"):                      # left mangled by rocprofv3 (it cannot demangle _Float16 parameters): _Z<len><name>[I<template args>E]...
        import re
        m = re.match(r"_Z(\d+)", n)
        if m:
            ln = int(m.group(1)); base = n[m.end():m.end() + ln]; rest = n[m.end() + ln:]
            t = re.match(r"I((?:L[bi]\d+E)+)E", rest)
            targs = "<" + ",".join(x[1:] for x in re.findall(r"L([bi]\d+)E", t.group(1))) + ">" if t else ""
            return base + targs
    return n.split("(")[0]


# 1. kernel trace (rocprofv3 --kernel-trace --stats): per (kernel, grid)
trace = os.path.join(SRC, "stats", "bench_kernel_trace.csv")
if os.path.exists(trace):
    d = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(trace)):
        k = short(r["Kernel_Name"])
        if not k.startswith("pn_"):
            continue
        key = (k[:90], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        d[key].append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[key] = (r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    d = {k: steady(v) for k, v in d.items()}
    tot = sum(sum(v) for v in d.values())
    with open(os.path.join(out, f"{tag}_kernel_stats{suf}.csv"), "w") as f:
        f.write(f"# kernels_snapshot={snap}\n")
        if a.frames_keep:
            f.write(f"# steady state: the last {a.frames_keep} of {a.frames_total} frames of the profiled run\n")
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us", "wg", "lds", "vgpr", "agpr", "sgpr", "scratch"])
        for key in sorted(d, key=lambda k: -sum(d[k])):
            v = d[key]
            w.writerow([key[0], key[1], len(v), f"{sum(v)/1e6:.3f}", f"{sum(v)/len(v)/1e3:.1f}", f"{100*sum(v)/tot:.2f}",
                        f"{min(v)/1e3:.1f}", f"{max(v)/1e3:.1f}", *meta[key]])

# 2. PMC passes (one counter set per run)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in sorted(os.listdir(SRC)) if os.path.isdir(SRC) else []:
    p = os.path.join(SRC, sub, "bench_counter_collection.csv")
    if not sub.startswith("pmc_") or not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if not k.startswith("pn_"):
            continue
        key = f"{k} grid={r['Grid_Size']}"
        agg[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
if agg:
    with open(os.path.join(out, f"{tag}_pmc_per_launch{suf}.csv"), "w") as f:
        f.write(f"# kernels_snapshot={snap}\n")
        if a.frames_keep:
            f.write(f"# steady state: the last {a.frames_keep} of {a.frames_total} frames of the profiled run\n")
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "avg", "min", "max"])
        for k in sorted(agg):
            for c in sorted(agg[k]):
                v = steady(agg[k][c])
                w.writerow([k, c, len(v), f"{sum(v)/len(v):.6g}", f"{min(v):.6g}", f"{max(v):.6g}"])
bj = os.path.join(SRC, "stats_bench.json")
if os.path.exists(bj):
    txt = open(bj).read().strip()
    if txt:
        open(os.path.join(out, f"{tag}_bench_under_rocprof{suf}.json"), "w").write(txt + "\n")
print("wrote", sorted(x for x in os.listdir(out) if x.startswith(tag)))
