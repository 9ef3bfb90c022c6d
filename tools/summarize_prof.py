"""Condense rocprofv3 outputs (gpurun_out/prof/*) into small tracked summaries under profiles/.

    python tools/summarize_prof.py <round-tag>        e.g. r01a
"""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]

# 1. kernel-trace stats (rocprofv3 --kernel-trace --stats)
rows = list(csv.DictReader(open(os.path.join(SRC, "stats", "bench_kernel_stats.csv"))))
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
    for r in rows:
        w.writerow([short(r["Name"])[:90], r["Calls"], f"{float(r['TotalDurationNs'])/1e6:.3f}",
                    f"{float(r['AverageNs'])/1e3:.1f}", r["Percentage"], f"{float(r['MinNs'])/1e3:.1f}",
                    f"{float(r['MaxNs'])/1e3:.1f}"])

# 2. PMC passes (one counter set per run)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(SRC, sub, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if not k.startswith("pn_"):
            continue
        key = f"{k} grid={r['Grid_Size']}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, f"{tag}_pmc_per_launch.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "avg", "min", "max"])
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            w.writerow([k, c, len(v), f"{sum(v)/len(v):.6g}", f"{min(v):.6g}", f"{max(v):.6g}"])
bj = os.path.join(SRC, "stats_bench.json")
if os.path.exists(bj):
    txt = open(bj).read().strip()
    if txt:
        open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(txt + "\n")
print("wrote", [x for x in os.listdir(out) if x.startswith(tag)])
