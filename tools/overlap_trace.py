"""Read a rocprofv3 kernel trace of tools/two_ctx.py and report how much of the front-end kernels' time overlaps network kernels
of the other stream.  usage: overlap_trace.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("pn_", "void pn_"))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], "fe" if ("frontend" in r["Kernel_Name"] or "pn_fe_" in r["Kernel_Name"]) else ("be" if "backend" in r["Kernel_Name"] else "nn")) for r in rows]
ev.sort()
t0 = ev[len(ev) // 2][0]
tot_fe = ov = 0
for s, e, st, k in ev:
    if k != "fe" or s < t0: continue
    tot_fe += e - s
    for s2, e2, st2, k2 in ev:
        if k2 == "nn" and st2 != st and e2 > s and s2 < e:
            ov += min(e, e2) - max(s, s2)
print(f"front-end kernel time {tot_fe/1e6:.2f} ms, of which concurrent with another stream's network kernels: {ov/1e6:.2f} ms ({100*ov/max(tot_fe,1):.0f} %)")
# a short timeline
names = {(int(r["Start_Timestamp"]), r["Stream_Id"]): r["Kernel_Name"].replace("void ", "").split("(")[0][:28] for r in rows}
for s, e, st, k in [x for x in ev if x[0] >= t0][:60]:
    print(f"  {(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  stream {st}  {k}  {names.get((s, st), '')}")
